"""A/B of the fused front end inside one process: per-layer HIP-event times (ms) with fusion on / off, interleaved."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "yolo-compression-and-deployment-in-fpga_amd")]
import numpy as np, torch
from yolo355 import prep, synth, _ffi
from yolo355.engine import Engine
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
u8 = len(sys.argv) > 2 and sys.argv[2] == "u8"
dev = torch.device("cuda", 0)
e = Engine([416, 416], 2, synth.ANCHOR_SIZE_MASK, conf_thresh=0.01, nms_thresh=0.5, max_batch=B, device=dev)
e.load_quantized(bench.quantized_layers(2))
sa = e.calibrate(synth.make_images(1, 1, 416, 416), [prep.RangeTracker() for _ in range(11)])
x = torch.from_numpy(synth.make_images(1000, B, 416, 416)).to(dev)
fr = torch.from_numpy(synth.make_frames_u8(1000, B, 416, 416)).to(dev)
e.profile(True)
res = {0: [], 1: []}
for rnd in range(6):
    for fuse in (1, 0):
        e.set_option(_ffi.OPT_FUSE_FRONT, fuse)
        acc = np.zeros(12)
        for i in range(10):
            if u8: e.forward_frames_device(fr)
            else: e.forward_device(x)
            acc += np.array(e.profile_ms())
        if rnd: res[fuse].append(acc / 10)
for fuse in (1, 0):
    m = np.median(np.array(res[fuse]), axis=0) * 1e3
    print("fuse=%d  conv1+conv2 %.1f us  (slots %s)  all conv %.1f  head %.1f nms %.1f" % (
        fuse, m[0] + m[1], np.round(m[:10], 1).tolist(), m[:10].sum(), m[10], m[11]))
