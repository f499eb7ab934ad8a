# s_memtime stamps of the fused front-end kernel (diagnostic build: make -C csrc EXTRA=-DFRONT_DIAG=1)
# per tile: 0 top, 1 loads landed, 2 Q done, 3 after B1, 4 C1 done, 5 after B2, 6 C2 done, 7 after B3 ; next tile's 0 = OUT done
import sys, os, numpy as np, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"yolo-compression-and-deployment-in-fpga_amd")]
import torch
from yolo355 import synth, prep, _ffi
from yolo355.engine import Engine
import bench
B=64
u8 = len(sys.argv) > 1 and sys.argv[1] == "u8"
eng = Engine([416,416], 2, synth.ANCHOR_SIZE_MASK, max_batch=B)
eng.load_quantized(bench.quantized_layers(2))
eng.calibrate(synth.make_images(1,1,416,416), [prep.RangeTracker() for _ in range(11)])
x = torch.from_numpy(synth.make_images(1000,B,416,416)).cuda()
fr = torch.from_numpy(synth.make_frames_u8(1000, B, 416, 416)).cuda()
lib=_ffi.lib()
lib.y355_debug_stamps.argtypes=[C.c_void_p, C.c_int, C.c_void_p, C.c_int]
np.set_printoptions(linewidth=250)
run = (lambda: eng.forward_frames_device(fr)) if u8 else (lambda: eng.forward_device(x))
for it in range(3): run()
lib.y355_debug_stamps(eng._h, 0, None, 0)
run(); eng.sync()
buf = np.zeros((1024,32), np.uint64)
lib.y355_debug_stamps(eng._h, -1, buf.ctypes.data, 1024)
t = buf.astype(np.int64)
names = ["load", "Q", "B1", "C1", "B2", "C2", "B3", "OUT"]
ok = t[:, 16] > 0
d = np.diff(t[ok][:, :17], axis=1)          # two tiles
print("workgroups with >= 2 tiles:", int(ok.sum()))
for k, n in enumerate(names):
    print("%-5s tile0 median %6d  tile1 median %6d   (p90 %6d)" % (n, np.median(d[:, k]), np.median(d[:, 8 + k]), np.percentile(d[:, 8 + k], 90)))
print("tile total median", int(np.median(t[ok][:, 8] - t[ok][:, 0])), int(np.median(t[ok][:, 16] - t[ok][:, 8])))
