# Three-handle regime: how long a step occupies its own stream (first event to last event of the step, profile mode 1) against
# the stream's period (3 x wall time per step): span ~ period means the stream is never idle between steps and the time goes
# into the launches and the gaps between them; span << period means the stream waits for its next step to arrive.
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
def run(n, prof):
    spans, names = [], None
    for e in engines: e.profile(prof)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        e = engines[i % NS]
        with torch.cuda.stream(streams[i % NS]):
            if prof and i >= 2 * NS: spans.append(e.profile_ms())
            e.forward_device(xs[i % 4], 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    for e in engines: e.profile(False)
    return dt, np.array(spans)
run(30, 0)
dt0, _ = run(150, 0)
dt1, sp = run(150, 1)
print("streams %d: wall per step %.1f us unprofiled, %.1f us with events; stream period %.1f us" % (NS, dt0 * 1e6, dt1 * 1e6, NS * dt1 * 1e6))
med = np.median(sp, axis=0) * 1e3
print("intervals between the step's events (us, median):", np.round(med, 1))
print("span of a step on its stream (sum): %.1f us = %.2f of the period" % (med.sum(), med.sum() / (NS * dt1 * 1e6)))
