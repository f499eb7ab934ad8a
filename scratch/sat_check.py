# how many outputs saturate per layer on the bench fixture (B=64 / 128), and what a forward costs then
import sys, os, numpy as np, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"yolo-compression-and-deployment-in-fpga_amd")]
import torch
from yolo355 import synth, prep
from yolo355.engine import Engine
import bench
for B in (64, 128):
    eng = Engine([416,416], 2, synth.ANCHOR_SIZE_MASK, max_batch=B)
    eng.load_quantized(bench.quantized_layers(2))
    sa = eng.calibrate(synth.make_images(1,1,416,416), [prep.RangeTracker() for _ in range(11)])
    for seed in (1000, 1001):
        x = torch.from_numpy(synth.make_images(seed,B,416,416)).cuda()
        for it in range(3): eng.forward_device(x)
        eng.sync(); t=time.time()
        for it in range(20): eng.forward_device(x)
        eng.sync(); dt=(time.time()-t)/20
        print("B", B, "seed", seed, "sat per layer", [eng.layer_stats(k)["saturated"] for k in range(10)], "counters", eng.counters(), "ms/step %.3f" % (dt*1e3))
    eng.close()
