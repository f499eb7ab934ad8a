"""soak of the product entry point: a Pipeline (three handles, throughput mode), N rounds of 6 tickets in flight (fp32 and uint8
routes alternating), every ticket's outputs compared bit for bit with a stand-alone Engine's; saturation counters too (the two
alternating counter sets of round 6).  usage: soak_pipeline.py [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "yolo-compression-and-deployment-in-fpga_amd")]
import numpy as np, torch
from yolo355 import prep, synth, _ffi
from yolo355.engine import Engine, Pipeline
import bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = 64
dev = torch.device("cuda", 0)
ql = bench.quantized_layers(2)
eng = Engine([416, 416], 2, synth.ANCHOR_SIZE_MASK, conf_thresh=0.01, nms_thresh=0.5, max_batch=B, device=dev)
eng.load_quantized(ql)
sa = eng.calibrate(synth.make_images(1, 1, 416, 416), [prep.RangeTracker() for _ in range(11)])
pipe = Pipeline([416, 416], 2, synth.ANCHOR_SIZE_MASK, conf_thresh=0.01, nms_thresh=0.5, max_batch=B, device=dev)
pipe.load_quantized(ql)
pipe.set_act_exponents(sa)
frames = [torch.from_numpy(synth.make_frames_u8(1000 + k, B, 416, 416)).to(dev) for k in range(2)]
xs = [torch.from_numpy(synth.normalize_frames(f.cpu().numpy())).to(dev) * (1.0 + 0.8 * k) for k, f in enumerate(frames)]   # the second one clamps
refs, sats = [], []
scratch = tuple(torch.zeros_like(t) for t in eng._buffers(B))
for k in range(2):
    for route in (0, 1):
        if route == 0:
            eng.forward_device(xs[k], 0, scratch)
        else:
            eng.forward_frames_device(frames[k], 0, scratch)
        torch.cuda.synchronize()
        refs.append([t.clone() for t in scratch])
        sats.append(eng.counters()[0])
print("reference saturation counts (fp32 x1, u8 x1, fp32 x1.8, u8):", sats)
bufs = [tuple(torch.zeros_like(t) for t in eng._buffers(B)) for _ in range(6)]
bad = 0
t0 = time.time()
last = {}                                                 # handle -> reference index of the last forward it ran
for it in range(rounds):
    what = []
    for i in range(6):
        k, route = (it + i) % 2, (it // 2 + i) % 2
        what.append(2 * k + route)
        t = pipe.submit(frames[k] if route else xs[k], 0, out=bufs[i], frames=bool(route), ordered=False)
        last[t % pipe.handles] = 2 * k + route
    pipe.sync()
    for i in range(6):
        r = refs[what[i]]
        n = r[3]
        ok = torch.equal(n, bufs[i][3])
        if ok:
            md = r[1].shape[1]
            m = torch.arange(md, device=dev)[None, :] < n[:, None].to(torch.int64)
            ok = torch.equal(r[1][m], bufs[i][1][m]) and torch.equal(r[2][m], bufs[i][2][m]) and torch.equal(r[0][m], bufs[i][0][m])
        if not ok:
            bad += 1
            print("MISMATCH round", it, "ticket", i)
    if it % 50 == 0:
        got = pipe.counters()[0]                          # summed over the handles' LAST forwards
        want = sum(sats[v] for v in last.values())
        if got != want:
            bad += 1
            print("COUNTER MISMATCH round", it, got, want)
print("soak_pipeline: %d rounds x 6 tickets, %d mismatches, %.1f s" % (rounds, bad, time.time() - t0))
sys.exit(1 if bad else 0)
