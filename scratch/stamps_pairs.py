# pairs_kernel phase stamps and per-wave trip counts (library built with EXTRA=-DY355_EXPERIMENTS, env Y355_NMS_STAMPS=1):
#   stamps_pairs.py slim_int8|slim_fp32|tiny_int8
import sys, os, numpy as np, ctypes as C
os.environ["Y355_NMS_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "yolo-compression-and-deployment-in-fpga_amd")]
import torch, bench
from yolo355 import synth, prep, _ffi
wl = sys.argv[1] if len(sys.argv) > 1 else "tiny_int8"
lib = _ffi.lib()
if wl == "slim_int8":
    from yolo355.engine import Engine
    B = 64
    eng = Engine([416, 416], 2, synth.ANCHOR_SIZE_MASK, conf_thresh=0.01, nms_thresh=0.5, max_batch=B)
    eng.load_quantized(bench.quantized_layers(2))
    eng.calibrate(synth.make_images(1, 1, 416, 416), [prep.RangeTracker() for _ in range(11)])
    x = torch.from_numpy(synth.make_images(1000, B, 416, 416)).cuda()
    for _ in range(3): eng.forward_device(x)
    eng.sync()
else:
    from yolo355.netengine import Net
    arch = "slim_yolo_v2" if wl == "slim_fp32" else "tiny_yolo_v3"
    classes = 2 if arch == "slim_yolo_v2" else 20
    B = 64 if arch == "slim_yolo_v2" else 128
    anchors = synth.ANCHOR_SIZE_MASK if arch == "slim_yolo_v2" else synth.TINY_MULTI_ANCHOR_SIZE
    A = len(anchors) if arch == "slim_yolo_v2" else len(anchors) // 2
    layers = synth.make_fp32_model(arch, 5, classes, A, pred_gain=1.5, obj_bias=-2.0)
    net = Net(arch, [416, 416], classes, anchors, 0.01, 0.5, max_batch=B, device="cuda:0", dtype="bf16")
    for i, L in enumerate(layers):
        w, b = L["w"].astype(np.float64), L["b"].astype(np.float64)
        if L["bn"] is not None:
            g, be, mu, var = (a.astype(np.float64) for a in L["bn"])
            sc = g / np.sqrt(var + 1e-5)
            w, b = w * sc[:, None, None, None], (b - mu) * sc + be
        net.load_layer(i, w.astype(np.float32), b.astype(np.float32))
    x = torch.from_numpy(synth.make_images(1000, B, 416, 416)).cuda()
    for _ in range(3): net.forward_device(x)
    net.sync()
buf = np.zeros(8 * 256 * 4, np.uint64)
lib.y355_debug_nms_stamps.argtypes = [C.c_void_p]
_ffi.check(lib.y355_debug_nms_stamps(buf.ctypes.data))
st = buf.reshape(4, 256, 8).astype(np.int64)
for k, name in ((0, "head_kernel"), (1, "pairs_kernel"), (2, "resolve_emit_kernel")):
    s = st[k]
    ok = s[:, 0] > 0
    if not ok.any(): continue
    t0 = s[ok, 0].min()
    rel = (s[ok] - t0).astype(np.float64)
    rel[s[ok] == 0] = np.nan
    print("%-20s workgroups stamped %3d; stamps (k cycles since the first start), median: %s ; end p50 %.1f p90 %.1f max %.1f" % (
        name, ok.sum(), np.round(np.nanmedian(rel, axis=0) / 1e3, 1).tolist(), np.nanmedian(rel[:, 7]) / 1e3, np.nanpercentile(rel[:, 7], 90) / 1e3, np.nanmax(rel[:, 7]) / 1e3))
w = st[3].reshape(-1)
t = w & ((1 << 40) - 1); trips = w >> 40
okw = t > 0
if okw.any():
    t0 = st[1][st[1][:, 0] > 0, 0].min() & ((1 << 40) - 1)
    rel = (t[okw] - t0) / 1e3
    print("pairs waves stamped %d: end of walk (k cycles) p10 %.1f p50 %.1f p90 %.1f max %.1f ; trips per wave p10 %d p50 %d p90 %d max %d ; cycles per trip (p50 wave) %.0f" % (
        okw.sum(), np.percentile(rel, 10), np.median(rel), np.percentile(rel, 90), rel.max(), np.percentile(trips[okw], 10), np.median(trips[okw]),
        np.percentile(trips[okw], 90), trips[okw].max(), 1e3 * np.median(rel) / max(1, np.median(trips[okw]))))
