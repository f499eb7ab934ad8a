# per-wave s_memrealtime stamps of convpx_kernel (-DPX_DIAG=2): 128 workgroups x 8 waves x 16 stamps
# per chunk: top, own wait done, past barrier, rows issued, groups done
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
np.set_printoptions(linewidth=250, suppress=True)
for it in range(3): eng.forward_device(x)
lib.y355_debug_stamps(eng._h, int(sys.argv[1]), None, 0)
eng.forward_device(x); eng.sync()
buf = np.zeros((1024,32), np.uint64)
lib.y355_debug_stamps(eng._h, -1, buf.ctypes.data, 1024)
t = buf.astype(np.int64).reshape(-1)[:128*8*16].reshape(128, 8, 16)
t0 = t[:, :, 0].min()
rel = (t - t0) / 100.0
print("median over workgroups, per wave (rows) x stamp (columns), us:")
print(np.round(np.median(rel, axis=0), 2))
