import sys, os, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"yolo-compression-and-deployment-in-fpga_amd"), os.path.join(ROOT,"tests")]
from oracle import yolo_oracle as O
from yolo355 import synth
from yolo355.engine import Engine
from yolo355.prep import RangeTracker
from cases import E2E
from helpers import dets_match
g = {}
for f in ("e2e.npz",):
    with np.load(os.path.join(ROOT,"tests","golden",f)) as z:
        for k in z.files: g[k]=z[k]
tag="gap"; wkw, anchors, pattern = E2E[tag]
H,W,C,cs = [int(v) for v in g[tag+"/meta"][:4]]; seeds=[int(v) for v in g[tag+"/meta"][4:]]
ql = O.quantize_layers(synth.make_weights(**wkw, num_classes=C))
eng = Engine([H,W], C, anchors, conf_thresh=0.01, max_batch=2); eng.load_quantized(ql)
xc = synth.make_images(cs,1,H,W,pattern)
eng.calibrate(xc,[RangeTracker() for _ in range(11)])
otr=[O.RangeTracker() for _ in range(11)]
O.detect(xc, ql, otr, [H,W], anchors, C, 0.01, 0.5)
xs = synth.make_images(seeds[0],1,H,W,pattern)
d = eng.forward(xs, tap=True)[0]
rb = O.detect(xs, ql, otr, [H,W], anchors, C, 0.01, 0.5, saturate=True)
r = rb["dets"][0]
print(len(d[1]), len(r[1]))
print("cls eq", np.array_equal(d[2], r[2]), "box maxdiff", np.abs(d[0]-r[0]).max(), "score maxdiff", np.abs(d[1]-r[1]).max())
i = np.argmax(np.abs(d[1]-r[1])); print(i, d[1][i], r[1][i], d[0][i], r[0][i], d[2][i], r[2][i])
cb, cs_, cc = eng.candidates(1)
print("cand score maxdiff", np.abs(cs_[0]-rb["cls_scores"][0].max(1)).max(), "cls mismatch", (cc[0]!=rb["cls_scores"][0].argmax(1)).sum())
