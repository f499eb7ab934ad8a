# s_memrealtime stamps (100 MHz) of conv3x3_i8_ring_kernel (diagnostic build: -DY355_DIAG=3) with the HW_ID / XCC_ID of every wave:
# which workgroups share a CU, how far apart their k-loops start, phase lengths per wave
# usage: stamps_ring_pairs.py [layer index = 7 (conv6)]
import sys, os, numpy as np, ctypes as C, collections
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
x = torch.from_numpy(synth.make_images(1000,B,416,416)).cuda()
lib=_ffi.lib()
lib.y355_debug_stamps.argtypes=[C.c_void_p, C.c_int, C.c_void_p, C.c_int]
np.set_printoptions(linewidth=250)
layer = int(sys.argv[1]) if len(sys.argv) > 1 else 7
ROWS = 4096
import time
t_end = time.time() + float(os.environ.get('STAMP_WARM_S', '2.5'))
while time.time() < t_end:
    for it in range(20): eng.forward_device(x)
    eng.sync()
lib.y355_debug_stamps(eng._h, layer, None, 0)
eng.forward_device(x); eng.sync()
buf = np.zeros((ROWS,32), np.uint64)
lib.y355_debug_stamps(eng._h, -1, buf.ctypes.data, ROWS)
t = buf.astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
rel = (t[:, :9] - t0) / 100.0
rel[t[:, :9] == 0] = 0
hw, xcc, ident = t[:, 9], t[:, 10] & 0xf, t[:, 11]
wg, wave = ident >> 4, ident & 15
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
print("layer", layer, "waves stamped:", len(t), "workgroups:", len(set(wg.tolist())), " span (us): %.2f" % rel.max())
d = np.diff(rel[:, :6], axis=1)
print("median phase lengths (us): prologue issue %.2f | first data %.2f | k-loop %.2f | epilogue %.2f | store drain %.2f" % tuple(np.median(d, axis=0)))
print("end of the last wave (us): %.2f" % rel[:, 5].max())
cyc = (t[:, 13] - t[:, 12]).astype(np.float64); us = (t[:, 3] - t[:, 2]) / 100.0
ok = (t[:, 12] > 0) & (us > 0)
if ok.any(): print("k-loop: median %.0f shader cycles in %.2f us -> in-kernel clock %.3f GHz (p10 %.3f p90 %.3f)" % (np.median(cyc[ok]), np.median(us[ok]), np.median(cyc[ok] / us[ok]) / 1e3, np.percentile(cyc[ok] / us[ok], 10) / 1e3, np.percentile(cyc[ok] / us[ok], 90) / 1e3))
# SIMD partners: which waves of a workgroup share a SIMD
by_wg = collections.defaultdict(list)
for i in range(len(t)): by_wg[int(wg[i])].append((int(wave[i]), int(simd[i])))
print("wave -> SIMD of the first workgroups:", [sorted(by_wg[k]) for k in sorted(by_wg)[:3]])
# workgroups per CU
cus = collections.defaultdict(set)
for i in range(len(t)): cus[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))].add(int(wg[i]))
cnt = collections.Counter(len(v) for v in cus.values())
print("CUs seen:", len(cus), " workgroups per CU histogram:", dict(cnt))
first = {}
for i in range(len(t)):
    w = int(wg[i])
    if w not in first: first[w] = [rel[i, 0], rel[i, 2], rel[i, 3]]
    else:
        first[w][0] = min(first[w][0], rel[i, 0]); first[w][1] = min(first[w][1], rel[i, 2]); first[w][2] = max(first[w][2], rel[i, 3])
pairs = [sorted(v) for v in cus.values() if len(v) == 2]
if pairs:
    print("examples of co-resident workgroups:", pairs[:8])
    dk = np.array([first[b][1] - first[a][1] for a, b in pairs])
    de = np.array([first[b][2] - first[a][2] for a, b in pairs])
    print("k-loop start offset between co-resident workgroups (us): median |d| %.3f p90 %.3f ; k-loop end offset median |d| %.3f p90 %.3f" % (
        np.median(np.abs(dk)), np.percentile(np.abs(dk), 90), np.median(np.abs(de)), np.percentile(np.abs(de), 90)))
