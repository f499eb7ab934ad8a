# candidates / suppressing pairs per image of a generic-net workload's NMS (y355_net_debug_nms): nms_stats.py slim_fp32|tiny_int8|tiny_bf16
import sys, os, numpy as np, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "yolo-compression-and-deployment-in-fpga_amd")]
import torch, bench
from yolo355 import synth, prep, _ffi
from yolo355.netengine import Net
wl = sys.argv[1] if len(sys.argv) > 1 else "slim_fp32"
arch = "slim_yolo_v2" if wl == "slim_fp32" else "tiny_yolo_v3"
classes = 2 if arch == "slim_yolo_v2" else 20
B = 64 if arch == "slim_yolo_v2" else 128
anchors = synth.ANCHOR_SIZE_MASK if arch == "slim_yolo_v2" else synth.TINY_MULTI_ANCHOR_SIZE
A = len(anchors) if arch == "slim_yolo_v2" else len(anchors) // 2
layers = synth.make_fp32_model(arch, 5, classes, A, pred_gain=1.5, obj_bias=-2.0)
folded = []
for L in layers:
    w, b = L["w"].astype(np.float64), L["b"].astype(np.float64)
    if L["bn"] is not None:
        g, be, mu, var = (a.astype(np.float64) for a in L["bn"])
        sc = g / np.sqrt(var + 1e-5)
        w, b = w * sc[:, None, None, None], (b - mu) * sc + be
    folded.append((w.astype(np.float32), b.astype(np.float32)))
net = Net(arch, [416, 416], classes, anchors, 0.01, 0.5, max_batch=B, device="cuda:0", dtype="bf16")
for i, (w, b) in enumerate(folded): net.load_layer(i, w, b)
x = torch.from_numpy(synth.make_images(1000, B, 416, 416)).cuda()
out = net.forward_device(x)
cnt = np.zeros(B, np.int32); ne = np.zeros(2 * B, np.int32)
_ffi.check(_ffi.lib().y355_net_debug_nms(net._h, B, cnt.ctypes.data, ne.ctypes.data))
print(wl, "candidates per image: min %d median %d max %d; edges listed: min %d median %d max %d; abandoned lists: %d of %d; detections per image %.0f" % (
    cnt.min(), np.median(cnt), cnt.max(), ne[0::2].min(), np.median(ne[0::2]), ne[0::2].max(), int((ne[1::2] != 0).sum()), B, out[3][:B].float().mean().item()))
