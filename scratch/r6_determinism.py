"""which tensor of the SlimYOLOv2 bf16 graph differs between two forwards of the same input (and between the thin-layer routes)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "yolo-compression-and-deployment-in-fpga_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from yolo355 import synth, _ffi
from yolo355.netengine import Net
B, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 2, 96, 160
if len(sys.argv) > 2: H = W = int(sys.argv[2])
layers = synth.make_fp32_model("slim_yolo_v2", 5, 2, 5, pred_gain=1.5, obj_bias=-2.0)
net = Net("slim_yolo_v2", [H, W], 2, synth.ANCHOR_SIZE_MASK, 0.01, 0.5, max_batch=B, device="cuda:0", dtype="bf16")
for i, L in enumerate(layers):
    w, b = L["w"].astype(np.float64), L["b"].astype(np.float64)
    if L["bn"] is not None:
        g, be, mu, var = (a.astype(np.float64) for a in L["bn"])
        sc = g / np.sqrt(var + 1e-5)
        w, b = w * sc[:, None, None, None], (b - mu) * sc + be
    net.load_layer(i, w.astype(np.float32), b.astype(np.float32))
x = synth.make_images(7, B, H, W)
for thin in (1, 0):
    net.set_option(_ffi.NET_OPT_THIN_RESIDENT, thin)
    for tap in (True, False):
        runs = []
        for rep in range(3):
            d = net.forward(x, tap=tap)
            t = [net.get_tensor(k, B).copy() for k in range(net.num_tensors)] if tap else []
            runs.append((d, t))
        for rep in (1, 2):
            same_d = all(np.array_equal(a, b) for i in range(B) for a, b in zip(runs[0][0][i], runs[rep][0][i]))
            diff_t = [k for k in range(len(runs[0][1])) if not np.array_equal(runs[0][1][k], runs[rep][1][k])]
            print("thin", thin, "tap", tap, "run", rep, "detections equal", same_d, "tensors that differ", diff_t)
