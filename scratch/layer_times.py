"""per-layer HIP-event times (us) of the library currently installed, one stream; optional 3-stream throughput.
usage: layer_times.py <name> <round> [thr]   |   layer_times.py --summary"""
import os, sys, json, time, glob, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "yolo-compression-and-deployment-in-fpga_amd")]
import numpy as np

if sys.argv[1] == "--summary":
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob("/tmp/lt_*.json")):
        d = json.load(open(f))
        acc[d["name"]].append(d)
    for n, rs in acc.items():
        ms = np.median(np.array([r["layer_us"] for r in rs]), axis=0)
        thr = np.median([r["thr3"] for r in rs]) if rs[0].get("thr3") else 0
        print("%-16s conv sum %6.1f | %s | head %5.1f nms %5.1f | 3-stream %8.0f img/s" % (
            n, ms[:10].sum(), " ".join("%5.1f" % v for v in ms[:10]), ms[10], ms[11], thr))
    sys.exit(0)

import torch
from yolo355 import prep, synth
from yolo355.engine import Engine
import bench
name, rnd = sys.argv[1], sys.argv[2]
thr = len(sys.argv) > 3 and sys.argv[3] == "thr"
B = 64
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(device=dev) for _ in range(3 if thr else 1)]
engs = []
for st in streams:
    with torch.cuda.stream(st):
        e = Engine([416, 416], 2, synth.ANCHOR_SIZE_MASK, conf_thresh=0.01, nms_thresh=0.5, max_batch=B, device=dev)
        e.load_quantized(bench.quantized_layers(2))
    engs.append(e)
with torch.cuda.stream(streams[0]):
    sa = engs[0].calibrate(synth.make_images(1, 1, 416, 416), [prep.RangeTracker() for _ in range(11)])
for e in engs:
    e.set_act_exponents(sa)
x = torch.from_numpy(synth.make_images(1000, B, 416, 416)).to(dev)
e = engs[0]
with torch.cuda.stream(streams[0]):
    for i in range(5):
        e.forward_device(x)
    e.profile(True)
    acc = []
    for i in range(20):
        e.forward_device(x)
        acc.append(e.profile_ms())
    e.profile(False)
res = dict(name=name, layer_us=(np.median(np.array(acc), axis=0) * 1e3).tolist())
if thr:
    bufs = [tuple(torch.empty_like(t) for t in e._buffers(B)) for _ in range(6)]

    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % 3]):
                engs[i % 3].forward_device(x, 0, bufs[i % 6])
    run(10)
    torch.cuda.synchronize()
    ts = []
    for r in range(7):
        t0 = time.perf_counter()
        run(60)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    res["thr3"] = B * 60 / float(np.median(ts))
json.dump(res, open("/tmp/lt_%s_%s.json" % (name, rnd), "w"))
