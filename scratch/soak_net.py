"""soak of the generic nets (convr.hip / convg.hip under three handles): N rounds of 6 interleaved steps of one workload, every
output compared bit for bit with the stand-alone result.  usage: soak_net.py tiny_int8|slim_fp32 [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "yolo-compression-and-deployment-in-fpga_amd")]
import numpy as np, torch
from yolo355 import prep, synth
from yolo355.netengine import Net
wl = sys.argv[1] if len(sys.argv) > 1 else "tiny_int8"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 300
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
quant = sa_in = sa = None
if dtype == "int8":
    quant = prep.quantize_folded(folded)
    fnet = Net(arch, [H, W], classes, anchors, 0.01, 0.5, max_batch=B, device=dev, dtype="bf16")
    for i, (w, b) in enumerate(folded): fnet.load_layer(i, w, b)
    sa_in, sa = fnet.calibration_exponents(synth.make_images(1, 1, H, W))
    del fnet
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
nets = []
for st in streams:
    with torch.cuda.stream(st):
        net = Net(arch, [H, W], classes, anchors, 0.01, 0.5, max_batch=B, device=dev, dtype=dtype)
        if dtype == "int8":
            for i, q in enumerate(quant): net.load_layer_i8(i, q["q_w"], q["q_b"], q["e_w"], q["e_b"])
            net.set_act_exponents(sa_in, sa)
        else:
            for i, (w, b) in enumerate(folded): net.load_layer(i, w, b)
    nets.append(net)
xs = [torch.from_numpy(synth.make_images(1000 + k, B, H, W)).to(dev) for k in range(2)]
refs = []
scratch = tuple(torch.zeros_like(t) for t in nets[0]._buffers(B))
for x in xs:
    nets[0].forward_device(x, 0, scratch)
    torch.cuda.synchronize()
    refs.append([t.clone() for t in scratch])
bufs = [tuple(torch.zeros_like(t) for t in nets[0]._buffers(B)) for _ in range(6)]
bad = 0
t0 = time.time()
for it in range(rounds):
    for i in range(6):
        with torch.cuda.stream(streams[i % 3]):
            nets[i % 3].forward_device(xs[(it + i) % 2], 0, bufs[i])
    torch.cuda.synchronize()
    for i in range(6):
        r = refs[(it + i) % 2]
        n = r[3]
        ok = torch.equal(n, bufs[i][3])
        if ok:
            md = r[1].shape[1]
            m = torch.arange(md, device=dev)[None, :] < n[:, None].to(torch.int64)
            ok = torch.equal(r[1][m], bufs[i][1][m]) and torch.equal(r[2][m], bufs[i][2][m]) and torch.equal(r[0][m], bufs[i][0][m])
        if not ok:
            bad += 1
            if bad < 10: print("MISMATCH round", it, "buffer", i)
print("soak_net %s: %d rounds x 6 steps, %d mismatches, %.1f s" % (wl, rounds, bad, time.time() - t0))
sys.exit(1 if bad else 0)
