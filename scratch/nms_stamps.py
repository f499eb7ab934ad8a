import sys, os, ctypes as C
os.environ["Y355_NMS_STAMPS"]="1"
sys.path.insert(0,'.'); sys.path.insert(0,'yolo-compression-and-deployment-in-fpga_amd')
import numpy as np, torch
import bench
from yolo355 import synth, prep, _ffi
from yolo355.engine import Engine
B=64
eng = Engine([416,416], 2, synth.ANCHOR_SIZE_MASK, 0.01, 0.5, max_batch=B, device="cuda:0")
eng.load_quantized(bench.quantized_layers(2))
sa = eng.calibrate(synth.make_images(1,1,416,416), [prep.RangeTracker() for _ in range(11)])
eng.set_act_exponents(sa)
x = torch.from_numpy(synth.make_images(1000,B,416,416)).cuda()
for _ in range(3): eng.forward_device(x)
torch.cuda.synchronize()
buf = np.zeros((4,256,8), np.uint64)
_ffi.check(_ffi.lib().y355_debug_nms_stamps(buf.ctypes.data))
names=["head","pairs","resolve_emit"]
for k in range(3):
    st = buf[k].astype(np.int64)
    act = st[:,0]>0
    st = st[act]
    t0 = st[:,0].min()
    print(names[k], "wgs", act.sum(), "start spread", (st[:,0].max()-t0), "end-of-kernel", (st[:,7].max()-t0))
    # per-phase median durations
    prev = st[:,0]
    for s in range(1,8):
        cur = st[:,s]
        ok = cur>0
        if ok.sum()==0: continue
        d = (cur-prev)[ok]
        print("   slot",s,"median %.0f max %.0f ticks (n=%d)"%(np.median(d), d.max(), ok.sum()))
        prev = np.where(ok, cur, prev)
# per-wave end-of-walk times (pairs, experiment build): block 3 = [wg*2 + wave/8][wave%8] of the first 128 workgroups
pw = buf[3].reshape(128, 16)
t1 = buf[1][:128, 1].astype(np.int64)            # staging done (wave 0)
for wg in range(0, 8):
    ends = (pw[wg] & np.uint64(0xffffffffff)).astype(np.int64) - (t1[wg] & 0xffffffffff)
    trips = (pw[wg] >> np.uint64(40)).astype(np.int64)
    print("pairs wg", wg, "per-wave walk cycles", ends.tolist(), "trips", trips.tolist())
