# per-wave timeline of the ring kernel (diagnostic build -DY355_DIAG=2): stamps of every wave of the first WGs
import sys, os, numpy as np, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"yolo-compression-and-deployment-in-fpga_amd")]
import torch
from yolo355 import synth, prep, _ffi
from yolo355.engine import Engine
import bench
B=64
eng = Engine([416,416], 2, synth.ANCHOR_SIZE_MASK, max_batch=B)
eng.load_quantized(bench.quantized_layers(2))
eng.calibrate(synth.make_images(1,1,416,416), [prep.RangeTracker() for _ in range(11)])
x = torch.from_numpy(synth.make_images(1000,B,416,416)).cuda()
lib=_ffi.lib()
lib.y355_debug_stamps.argtypes=[C.c_void_p, C.c_int, C.c_void_p, C.c_int]
np.set_printoptions(linewidth=250)
for layer in [int(a) for a in sys.argv[1:]]:
    for it in range(3): eng.forward_device(x)
    lib.y355_debug_stamps(eng._h, layer, None, 0)
    eng.forward_device(x); eng.sync()
    buf = np.zeros((1024,32), np.uint64)
    lib.y355_debug_stamps(eng._h, -1, buf.ctypes.data, 1024)
    t = buf.astype(np.int64).reshape(128, 8, 32)
    print("layer", layer)
    for wg in [0, 5, 77]:
        base = t[wg,0,0]
        print(" wg", wg, "start per wave", (t[wg,:,0]-base).tolist())
        # stamps 2.. : triples (before wait, after wait, after barrier) for the 9 steps of chunk 1
        for st in range(0, 8):
            a = t[wg,:,2+3*st]-base; b_ = t[wg,:,3+3*st]-base; c = t[wg,:,4+3*st]-base
            print("   step", st, "arrive", a.tolist(), "| waited", (b_-a).tolist(), "| release", (c-base*0).tolist())
