# s_memrealtime stamps (100 MHz) of conv3x3_i8_ring_kernel (diagnostic build: -DY355_DIAG=3), per wave:
# 0 entry | 1 prologue issued | 2 first data landed (past the first barrier) | 3 k-loop done | 4 epilogue's stores issued | 5 stores retired
# usage: stamps_ring.py [layer index = 7 (conv6)]
import sys, os, numpy as np, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"yolo-compression-and-deployment-in-fpga_amd")]
import torch
from yolo355 import synth, prep, _ffi
from yolo355.engine import Engine
import bench
B=int(os.environ.get('STAMP_B','64'))
eng = Engine([416,416], 2, synth.ANCHOR_SIZE_MASK, max_batch=B)
eng.load_quantized(bench.quantized_layers(2))
sa = eng.calibrate(synth.make_images(1,1,416,416), [prep.RangeTracker() for _ in range(11)])
print("exponents", sa)
x = torch.from_numpy(synth.make_images(1000,B,416,416)).cuda()
lib=_ffi.lib()
lib.y355_debug_stamps.argtypes=[C.c_void_p, C.c_int, C.c_void_p, C.c_int]
np.set_printoptions(linewidth=250)
layer = int(sys.argv[1]) if len(sys.argv) > 1 else 7
for it in range(3): eng.forward_device(x)
lib.y355_debug_stamps(eng._h, layer, None, 0)
eng.forward_device(x); eng.sync()
buf = np.zeros((1024,32), np.uint64)
lib.y355_debug_stamps(eng._h, -1, buf.ctypes.data, 1024)
t = buf.astype(np.int64)[:, :9]
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
rel = (t - t0) / 100.0
rel[t == 0] = 0
print("layer", layer, "waves stamped:", len(t), " span (us): %.2f" % rel.max())
for q in (10, 50, 90):
    print("p%d since the first wave's entry (us):" % q, np.round(np.percentile(rel, q, axis=0), 2).tolist())
d = np.diff(rel[:, :6], axis=1)
print("median phase lengths (us): prologue issue %.2f | first data %.2f | k-loop %.2f | epilogue %.2f | store drain %.2f" % tuple(np.median(d, axis=0)))
two = t[:, 6] > 0
if two.any():
    r2 = rel[two]
    print("waves with a second tile: %d; second tile (us): pre-phase %.2f | k-loop %.2f | epilogue %.2f" % (
        two.sum(), np.median(r2[:, 6] - r2[:, 4]), np.median(r2[:, 7] - r2[:, 6]), np.median(r2[:, 8] - r2[:, 7])))
