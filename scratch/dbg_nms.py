import sys, os, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"yolo-compression-and-deployment-in-fpga_amd"), os.path.join(ROOT,"tests")]
from oracle import yolo_oracle as O
from yolo355 import synth
from yolo355.engine import Engine
from yolo355.prep import RangeTracker
H=W=416
ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2))
eng = Engine([H,W], 2, synth.ANCHOR_SIZE_MASK, conf_thresh=0.01, max_batch=6)
eng.load_quantized(ql)
xc = synth.make_images(1,1,H,W)
print(eng.calibrate(xc, [RangeTracker() for _ in range(11)]))
for tap in (False, True, False):
    for find in (False, True):
        d = eng.forward(xc, find=find, tap=tap)
        print("tap", tap, "find", find, "ndet", len(d[0][1]), eng.counters())
eng.set_thresholds(0.1, 0.5)
d = eng.forward(xc); print("thr 0.1 ndet", len(d[0][1]))
eng.set_thresholds(0.01, 0.5)
d = eng.forward(xc, tap=True); print("thr 0.01 tap ndet", len(d[0][1]))
