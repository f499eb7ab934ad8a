"""host enqueue rate vs GPU rate of the 3-stream loop (is the headline host-bound?)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "yolo-compression-and-deployment-in-fpga_amd")]
import numpy as np, torch
from yolo355 import prep, synth
from yolo355.engine import Engine
import bench
B = 64
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
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
bufs = [tuple(torch.empty_like(t) for t in engs[0]._buffers(B)) for _ in range(6)]
def run(n):
    for i in range(n):
        with torch.cuda.stream(streams[i % 3]):
            engs[i % 3].forward_device(x, 0, bufs[i % 6])
run(12); torch.cuda.synchronize()
for n in (20, 60, 200):
    t0 = time.perf_counter(); run(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("steps %d: enqueue %.1f us/step, total %.1f us/step" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
