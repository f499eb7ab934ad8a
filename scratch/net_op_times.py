# per-op HIP-event times (us) of a generic-net workload of bench.py (one stream): net_op_times.py tiny_int8|slim_fp32
import sys, os, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"yolo-compression-and-deployment-in-fpga_amd")]
import torch
import bench
from yolo355 import synth, prep
from yolo355.netengine import Net
wl = sys.argv[1] if len(sys.argv) > 1 else "tiny_int8"
arch = "slim_yolo_v2" if wl == "slim_fp32" else "tiny_yolo_v3"
dtype = "int8" if wl == "tiny_int8" else "bf16"
classes = 2 if arch == "slim_yolo_v2" else 20
B = 64 if arch == "slim_yolo_v2" else 128
anchors = synth.ANCHOR_SIZE_MASK if arch == "slim_yolo_v2" else synth.TINY_MULTI_ANCHOR_SIZE
A = len(anchors) if arch == "slim_yolo_v2" else len(anchors) // 2
H = W = 416
dev = torch.device("cuda", 0)
layers = synth.make_fp32_model(arch, 5, classes, A, pred_gain=1.5, obj_bias=-2.0)
folded = []
for L in layers:
    w, b = L["w"].astype(np.float64), L["b"].astype(np.float64)
    if L["bn"] is not None:
        g, be, mu, var = (a.astype(np.float64) for a in L["bn"])
        sc = g / np.sqrt(var + 1e-5)
        w, b = w * sc[:, None, None, None], (b - mu) * sc + be
    folded.append((w.astype(np.float32), b.astype(np.float32)))
net = Net(arch, [H, W], classes, anchors, 0.01, 0.5, max_batch=B, device=dev, dtype=dtype)
if dtype == "int8":
    quant = prep.quantize_folded(folded)
    fnet = Net(arch, [H, W], classes, anchors, 0.01, 0.5, max_batch=B, device=dev, dtype="bf16")
    for i, (w, b) in enumerate(folded): fnet.load_layer(i, w, b)
    sa_in, sa = fnet.calibration_exponents(synth.make_images(1, 1, H, W))
    del fnet
    for i, q in enumerate(quant): net.load_layer_i8(i, q["q_w"], q["q_b"], q["e_w"], q["e_b"])
    net.set_act_exponents(sa_in, sa)
else:
    for i, (w, b) in enumerate(folded): net.load_layer(i, w, b)
x = torch.from_numpy(synth.make_images(1000, B, H, W)).to(dev)
for _ in range(3): net.forward_device(x)
net.profile(True)
acc = []
for _ in range(10):
    net.forward_device(x); acc.append(np.array(net.profile_ms()))
net.profile(False)
ms = np.median(np.array(acc), axis=0) * 1e3
print(wl, "per-op us:", " ".join("%d:%.1f" % (i, v) for i, v in enumerate(ms)), "| sum %.1f" % ms.sum())
