# One handle alone with the throughput-mode grids (Y355_OPT_RING_WORKGROUPS = 128): the launches' own durations and the CU-time
# (CUs a grid holds x duration) of a step, against 256 CUs x the three-handle step time.
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "yolo-compression-and-deployment-in-fpga_amd")]
import torch
from yolo355 import synth, prep
from yolo355.engine import Engine
import bench
B = 64
eng = Engine([416, 416], 2, synth.ANCHOR_SIZE_MASK, max_batch=B)
eng.load_quantized(bench.quantized_layers(2))
eng.calibrate(synth.make_images(1, 1, 416, 416), [prep.RangeTracker() for _ in range(11)])
xs = [torch.from_numpy(synth.make_images(1000 + i, B, 416, 416)).cuda() for i in range(4)]
names = ["front", "-", "conv3_1", "conv3_2", "conv4_1", "conv4_2", "conv5", "conv6", "conv7", "pred", "decode", "head", "pairs", "resolve"]
for wgs, cus in ((0, [256, 0, 256, 256, 256, 256, 256, 256, 256, 256, 112, 32, 128, 64]), (128, [256, 0, 128, 128, 128, 128, 128, 128, 128, 128, 112, 32, 64, 64])):
    eng.set_option(2, wgs)
    for i in range(20): eng.forward_device(xs[i % 4], 0)
    eng.profile(2)
    acc = []
    for i in range(30):
        eng.forward_device(xs[i % 4], 0)
        acc.append(eng.profile_kernels_ms())
    eng.profile(False)
    k = np.median(np.array(acc), axis=0) * 1e3
    print("option %3d: " % wgs + "  ".join("%s %.1f" % (n, v) for n, v in zip(names, k) if n != "-"))
    print("   sum %.1f us; CU-time (grid CUs x duration) %.0f CU-us = %.1f us of a 256-CU GPU" % (k.sum(), float((k * np.array(cus)).sum()), float((k * np.array(cus)).sum()) / 256))
