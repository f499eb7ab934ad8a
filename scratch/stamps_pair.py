# s_memrealtime stamps (100 MHz) of pxpair3_kernel (variant built with -DPAIR_DIAG=1), thread 0 of every workgroup:
# 0 entry | 1 weights + first rows landed | per step: phase A done (this wave), past barrier, phase B done (this wave), past barrier | last: drained
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
if len(sys.argv) > 1: eng.set_option(_ffi.OPT_RING_WORKGROUPS, int(sys.argv[1]))
if len(sys.argv) > 2: eng.set_option(_ffi.OPT_FUSE_PAIRS, int(sys.argv[2]))
x = torch.from_numpy(synth.make_images(1000,B,416,416)).cuda()
lib=_ffi.lib()
lib.y355_debug_stamps.argtypes=[C.c_void_p, C.c_int, C.c_void_p, C.c_int]
np.set_printoptions(linewidth=250)
for it in range(3): eng.forward_device(x)
LAYER = int(sys.argv[3]) if len(sys.argv) > 3 else 2      # 2: conv3 pair, 4: conv4 pair
lib.y355_debug_stamps(eng._h, LAYER, None, 0)
eng.forward_device(x); eng.sync()
buf = np.zeros((1024,32), np.uint64)
lib.y355_debug_stamps(eng._h, -1, buf.ctypes.data, 1024)
t = buf.astype(np.int64)[:256]
t = t[t[:, 0] > 0]
if len(sys.argv) <= 2 or sys.argv[2] == "1":     # role-split kernel (the default): stamps 0-15 of thread 0 (a conv3_1 wave), 16-31 of thread 256 (a conv3_2 wave)
    t0 = t[:, 0].min()
    for name, tt in (("conv3_1 wave 0", t[:, :16]), ("conv3_2 wave 4", t[:, 16:])):
        n = int((tt[0] > 0).sum())
        d = np.diff(tt[:, :n], axis=1) / 100.0
        print(name, "stamps", n, "median intervals (us):", np.round(np.median(d, axis=0), 2).tolist(), "last stamp since entry p50 %.2f" % (np.median(tt[:, n - 1] - t[:, 0]) / 100))
    sys.exit(0)
n = int((t[0] > 0).sum())
t0 = t[:, 0].min()
print("workgroups", len(t), "stamps per workgroup:", n, " kernel span (us): %.2f" % ((t[:, :n].max() - t0) / 100.0))
d = np.diff(t[:, :n], axis=1) / 100.0
print("median interval between consecutive stamps (us):", np.round(np.median(d, axis=0), 2).tolist())
print("entry since first (us) p50 %.2f p90 %.2f; own duration p50 %.2f max %.2f" % (np.median(t[:,0]-t0)/100, np.percentile(t[:,0]-t0, 90)/100, np.median(t[:,n-1]-t[:,0])/100, (t[:,n-1]-t[:,0]).max()/100))
