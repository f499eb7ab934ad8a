# Three handles: does the order in which the host hands steps to the streams matter?  pattern k: stream = (i // k) % 3
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "yolo-compression-and-deployment-in-fpga_amd")]
import torch
from yolo355 import synth, prep
from yolo355.engine import Engine
import bench
B, NS = 64, int(os.environ.get("STREAMS", "3"))
dev = torch.device("cuda:0")
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
engines = []
for st in streams:
    with torch.cuda.stream(st):
        e = Engine([416, 416], 2, synth.ANCHOR_SIZE_MASK, max_batch=B)
        e.load_quantized(bench.quantized_layers(2))
        e.calibrate(synth.make_images(1, 1, 416, 416), [prep.RangeTracker() for _ in range(11)])
        engines.append(e)
xs = [torch.from_numpy(synth.make_images(1000 + i, B, 416, 416)).cuda() for i in range(4)]
torch.cuda.synchronize()
for e in engines:
    e.set_option(2, 128 if NS > 1 else 0)
def run(n, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        j = (i // k) % NS
        with torch.cuda.stream(streams[j]):
            engines[j].forward_device(xs[i % 4], 0)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, th / n
run(60, 1)
for rep in range(3):
    for k in (1, 2, 4, 8):
        dt, th = run(240, k)
        print("round %d: %d consecutive steps per stream: %.1f us per step (host submit %.1f us per step) = %.0f img/s" % (rep, k, dt * 1e6, th * 1e6, B / dt))
