# s_memrealtime stamps (100 MHz) of convpx32_kernel (diagnostic build: -DPX_DIAG=1), per workgroup, thread 0:
# 0 entry | per tile: top, slab landed (own wait), past barrier, groups done | last: stores drained
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
for it in range(3): eng.forward_device(x)
lib.y355_debug_stamps(eng._h, int(sys.argv[1]) if len(sys.argv) > 1 else 2, None, 0)
eng.forward_device(x); eng.sync()
buf = np.zeros((1024,32), np.uint64)
lib.y355_debug_stamps(eng._h, -1, buf.ctypes.data, 1024)
t = buf.astype(np.int64)[:256]
n = int((t[0] > 0).sum())
t0 = t[:, 0].min()
print("stamps per workgroup:", n, " kernel span (us): %.2f" % ((t[:, :n].max() - t0) / 100.0))
rel = (t[:, :n] - t0) / 100.0
print("median time of each stamp since the first workgroup's entry (us):", np.round(np.median(rel, axis=0), 2).tolist())
print("p10:", np.round(np.percentile(rel, 10, axis=0), 2).tolist())
print("p90:", np.round(np.percentile(rel, 90, axis=0), 2).tolist())
