import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "yolo-compression-and-deployment-in-fpga_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from yolo355 import synth, _ffi
from oracle import yolo_oracle as O
from test_dropin import _model
cfg = dict(weights=dict(seed=2, pred_gain=400.0, obj_bias=-4.0), size=[240, 320], classes=20, seeds=[51, 52, 53, 54, 55], conf=0.1)
size, classes = cfg["size"], cfg["classes"]
x = np.concatenate([synth.make_images(s, 1, size[0], size[1], "blocks") for s in cfg["seeds"]])
r2 = np.load(os.path.join(ROOT, "tests/golden/r2.npz"))
ql = O.quantize_layers(synth.make_weights(**cfg["weights"], num_classes=classes))
otr = [O.RangeTracker() for _ in range(11)]
O.detect(x[:1], ql, otr, size, synth.ANCHOR_SIZE, classes, cfg["conf"], 0.5)
ro = O.detect(x, ql, otr, size, synth.ANCHOR_SIZE, classes, cfg["conf"], 0.5, saturate=True, keep=True)
print("oracle sat", ro["sat"])
for fuse in (1, 0):
    net = _model(synth.make_weights(**cfg["weights"], num_classes=classes), classes, synth.ANCHOR_SIZE, size, cfg["conf"], "cuda:0")
    xt = torch.from_numpy(x)
    net.forward_batch(xt[:1], quantization=True)
    net._engine.set_option(_ffi.OPT_FUSE_FRONT, fuse)
    for lo, hi in ((0, 2), (2, 4), (4, 5)):
        d = net.forward_batch(xt[lo:hi], quantization=True)
        eng = net._engine
        pred = eng.get_feature(9, hi - lo)
        print("fuse", fuse, "batch", lo, hi, "pred equal oracle:", np.array_equal(pred, ro["pred_q"][lo:hi].astype(np.int8)),
              "ndet", [len(a[1]) for a in d], "oracle", [len(ro["dets"][i][1]) for i in range(lo, hi)], "ctr", eng.counters())
    i = 4
    refn = sum(len(r2["eval/boxes/%d/%d" % (j, i)]) for j in range(classes))
    print("golden image 4 dets:", refn)
    for j in range(classes):
        a = r2["eval/boxes/%d/%d" % (j, i)]
        if len(a): print(" cls", j, a)
    print("ours", [(c, b * np.array([320, 240, 320, 240]), s) for b, s, c in zip(*d[0])])
    print("oracle", [(c, b * np.array([320, 240, 320, 240]), s) for b, s, c in zip(*ro["dets"][4][:3])])
