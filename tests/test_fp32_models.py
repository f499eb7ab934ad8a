"""fp32 model families: SlimYOLOv2 and YOLOv3tiny.

CPU: the torch fp32 oracle (oracle/fp32_oracle.py) against the goldens the reference's own classes
produced (tests/golden/fp32.npz), and the drop-in classes' parameter layout.
GPU: the bf16-MFMA engine (y355_net) against the oracle and the goldens.

Tolerances of the bf16 path (inputs, weights and inter-layer activations rounded to bf16, 8
significant bits, fp32 accumulation, fp32 prediction maps and head), measured over these cases:
  prediction maps   relative L2 error <= 1.5e-2 (measured 5e-3..7e-3), max |err| <= 2.5e-2 * max |pred|
  per-anchor scores |err| <= 0.04 everywhere (measured max 0.016)
  per-anchor boxes  |err| <= 0.03 (normalised units) on 98 % of the anchors, <= 0.08 everywhere
                    (exp(tw) amplifies the logit error; measured max 0.038)
  NMS               exact: the engine's detections equal the reference post-processing run on the
                    engine's own per-anchor decode
  detections vs the fp32 run: greedy NMS amplifies 1 % score changes among heavily overlapping
                    boxes, so final lists are compared loosely: >= 80 % of either side matched at
                    (same class, IoU >= 0.8, |score err| <= 0.05), >= 90 % at (IoU >= 0.5, 0.2);
                    counts within 3 %
"""
import numpy as np
import pytest
import torch

from cases import FP32_CASES, fp32_setup
from helpers import explain_detection_differences, dets_close, dets_match
from oracle import fp32_oracle as F
from oracle import yolo_oracle as O
from yolo355 import synth

IDS = [c[0] for c in FP32_CASES]


@pytest.fixture(scope="module")
def gold():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "fp32.npz"))


@pytest.mark.parametrize("case", FP32_CASES, ids=IDS)
def test_oracle_matches_reference(case, gold):
    """the fp32 restatement reproduces the reference's prediction maps and detections"""
    tag, arch, size, classes = case[:4]
    layers, anchors, A, x = fp32_setup(case)
    names = ["pred"] if arch == "slim_yolo_v2" else ["pred_1", "pred_2"]
    for conf in (0.01, 0.1):
        r = F.detect(arch, layers, x, size, anchors, classes, conf, 0.5)
        for n, p in zip(names, r["preds"]):
            g = gold["%s/%s" % (tag, n)]
            assert p.shape == g.shape
            # batch > 1 takes another conv blocking inside torch: a few ulp
            assert np.abs(p - g).max() <= 1e-4 * max(1.0, np.abs(g).max())
        for bi in range(x.shape[0]):
            ref = tuple(gold["%s/%d/det%g/%s" % (tag, bi, conf, k)] for k in ("boxes", "scores", "cls"))
            sc = r["cls_scores"][bi].max(axis=1)
            ok, msg = dets_match(ref, r["dets"][bi][:3], box_tol=1e-4, score_tol=1e-4, all_scores=sc)
            assert ok, "%s image %d conf %g: %s" % (tag, bi, conf, msg)


def _state_dict_of(case):
    tag, arch, size, classes = case[:4]
    layers, anchors, A, x = fp32_setup(case)
    return {k: torch.from_numpy(v.copy()) for k, v in synth.state_dict_fp32(layers).items()}, layers, anchors, x


@pytest.mark.parametrize("case", [FP32_CASES[2], FP32_CASES[4]], ids=[IDS[2], IDS[4]])
def test_dropin_state_dict_layout(case):
    """the drop-in classes take the reference's state_dict keys unchanged (strict load)"""
    from yolo355.models import SlimYOLOv2, YOLOv3tiny
    tag, arch, size, classes = case[:4]
    sd, layers, anchors, x = _state_dict_of(case)
    cls = SlimYOLOv2 if arch == "slim_yolo_v2" else YOLOv3tiny
    m = cls("cpu", input_size=size, num_classes=classes, anchor_size=anchors)
    full = m.state_dict()
    assert set(sd) <= set(full)
    assert all(k.endswith("num_batches_tracked") for k in set(full) - set(sd))
    m.load_state_dict(sd, strict=False)
    m.eval()
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):        # no GPU, no fallback
            m(torch.from_numpy(x))


def test_bn_fold_is_exact():
    """folded_f32 (W', b') equals conv -> BN(eval) of the un-fused block"""
    from yolo355.utils.modules import Conv2d, folded_f32
    torch.manual_seed(0)
    m = Conv2d(8, 12, 3, 1, leakyReLU=True).eval()
    with torch.no_grad():
        m.convs[1].running_mean.uniform_(-0.5, 0.5)
        m.convs[1].running_var.uniform_(0.5, 2.0)
        m.convs[1].weight.uniform_(0.5, 1.5)
        m.convs[1].bias.uniform_(-0.3, 0.3)
    x = torch.randn(2, 8, 9, 11)
    w, b = folded_f32(m.convs)
    with torch.no_grad():
        ref = m.convs[1](m.convs[0](x))
        got = torch.nn.functional.conv2d(x, torch.from_numpy(w), torch.from_numpy(b), padding=1)
    assert torch.allclose(ref, got, atol=2e-5, rtol=1e-5)


# ------------------------------------------------------------------------------------------ GPU
def _load_net(case, max_batch):
    from yolo355.netengine import Net
    from yolo355.utils.modules import folded_f32  # noqa: F401
    tag, arch, size, classes = case[:4]
    layers, anchors, A, x = fp32_setup(case)
    net = Net(arch, size, classes, anchors, 0.01, 0.5, max_batch=max_batch, device="cuda:0", dtype="bf16")
    for i, L in enumerate(layers):
        w, b = L["w"].astype(np.float64), L["b"].astype(np.float64)
        if L["bn"] is not None:
            g, be, mu, var = (a.astype(np.float64) for a in L["bn"])
            s = g / np.sqrt(var + F.EPS)
            w, b = w * s[:, None, None, None], (b - mu) * s + be
        net.load_layer(i, w.astype(np.float32), b.astype(np.float32))
    return net, layers, anchors, x


@pytest.mark.gpu
@pytest.mark.parametrize("case", FP32_CASES, ids=IDS)
def test_engine_vs_oracle(case, gold):
    tag, arch, size, classes = case[:4]
    net, layers, anchors, x = _load_net(case, len(case[5]))
    r = F.detect(arch, layers, x, size, anchors, classes, 0.01, 0.5)
    out = net.forward(x, tap=True)
    B = x.shape[0]
    # prediction maps (the last tensors of the graph)
    for k, p in enumerate(r["preds"]):
        got = net.get_tensor(net.num_tensors - len(r["preds"]) + k, B)
        assert got.shape == p.shape
        err = got.astype(np.float64) - p
        rel = np.sqrt((err ** 2).sum() / (p.astype(np.float64) ** 2).sum())
        assert rel <= 1.5e-2, "%s pred %d: relative L2 error %.3g" % (tag, k, rel)
        assert np.abs(err).max() <= 2.5e-2 * np.abs(p).max(), "%s pred %d: max err %.3g" % (tag, k, np.abs(err).max())
        # ... and the reference's own map
        gname = ["pred"] if arch == "slim_yolo_v2" else ["pred_1", "pred_2"]
        g = gold["%s/%s" % (tag, gname[k])]
        relg = np.sqrt(((got - g).astype(np.float64) ** 2).sum() / (g.astype(np.float64) ** 2).sum())
        assert relg <= 1.5e-2
    # inter-layer activations
    for k, t in enumerate(r["taps"]):
        got = net.get_tensor(k, B).astype(np.float64)
        rel = np.sqrt(((got - t) ** 2).sum() / (t.astype(np.float64) ** 2).sum())
        assert rel <= 1.5e-2, "%s tensor %d: relative L2 error %.3g" % (tag, k, rel)
    # per-anchor decode
    cb, cs, cc = net.candidates(B)
    best = r["cls_scores"].max(axis=2)
    assert np.abs(cs - best).max() <= 0.04
    db = np.abs(cb - r["box"]).max(axis=2)
    assert (db <= 0.03).mean() >= 0.98 and db.max() <= 0.08
    for bi in range(B):
        # NMS: exact on the engine's own decode
        prob = np.zeros((cs.shape[1], classes), np.float32)
        prob[np.arange(cs.shape[1]), cc[bi]] = cs[bi]
        own = O.postprocess(cb[bi], prob, 0.01, 0.5, classes)
        ok, msg = dets_match(own[:3], out[bi], box_tol=0, score_tol=0, all_scores=cs[bi])
        assert ok, "%s image %d NMS: %s" % (tag, bi, msg)
        # final lists against the fp32 oracle and the reference's golden
        for ref in (r["dets"][bi], tuple(gold["%s/%d/det0.01/%s" % (tag, bi, k)] for k in ("boxes", "scores", "cls"))):
            fr, fg = dets_close(ref, out[bi], 0.8, 0.05)
            assert fr >= 0.8 and fg >= 0.8, "%s image %d: matched %.3f of ref, %.3f of got" % (tag, bi, fr, fg)
            fr, fg = dets_close(ref, out[bi], 0.5, 0.2)
            assert fr >= 0.9 and fg >= 0.9, "%s image %d: loosely matched %.3f / %.3f" % (tag, bi, fr, fg)
            assert abs(len(out[bi][1]) - len(ref[1])) <= max(0.03 * len(ref[1]), 5)
        # every difference between the two lists has a per-anchor cause inside the tolerances above
        ex = explain_detection_differences(r["box"][bi], best[bi], r["cls_scores"][bi].argmax(axis=1), cb[bi], cs[bi], cc[bi], 0.01, 0.5)
        assert ex["n_got"] == len(out[bi][1]) and abs(ex["n_ref"] - len(r["dets"][bi][1])) <= 2
        assert ex["unexplained"] == [], "%s image %d: %d list differences without a per-anchor cause" % (tag, bi, len(ex["unexplained"]))
        assert ex["root_score_dev"] <= 0.04 and ex["root_box_dev"] <= 0.08, ex
    net.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [FP32_CASES[2], FP32_CASES[4]], ids=[IDS[2], IDS[4]])
def test_dropin_forward(case):
    """SlimYOLOv2 / YOLOv3tiny drop-ins: load_state_dict -> eval -> forward == engine, image 0 first"""
    from yolo355.models import SlimYOLOv2, YOLOv3tiny
    tag, arch, size, classes = case[:4]
    sd, layers, anchors, x = _state_dict_of(case)
    cls = SlimYOLOv2 if arch == "slim_yolo_v2" else YOLOv3tiny
    m = cls("cuda:0", input_size=size, num_classes=classes, anchor_size=anchors)
    m.load_state_dict(sd, strict=False)
    m.eval()
    b, s, c = m(torch.from_numpy(x))
    assert b.dtype == np.float32 and b.flags.writeable and c.dtype == np.int64
    allimg = m.forward_batch(torch.from_numpy(x))
    assert len(allimg) == x.shape[0] and np.array_equal(allimg[0][1], s)
    r = F.detect(arch, layers, x, size, anchors, classes, 0.01, 0.5)
    for bi in range(x.shape[0]):
        fr, fg = dets_close(r["dets"][bi], allimg[bi], 0.8, 0.05)
        assert fr >= 0.8 and fg >= 0.8
    # element i of a batch equals the image run alone (bit-exact: same kernels, same tiles per image)
    alone = m.forward_batch(torch.from_numpy(x[1:2]))[0]
    assert np.array_equal(alone[0], allimg[1][0]) and np.array_equal(alone[1], allimg[1][1])
    m.train()
    with pytest.raises(NotImplementedError):
        m(torch.from_numpy(x))


# ------------------------------------------------------------------ int8 form of YOLOv3tiny (C4)
TINY = [c for c in FP32_CASES if c[1] == "tiny_yolo_v3"]


def test_int8_oracle_tracks_fp32():
    """CPU: the integer restatement (exponents from the fp32 oracle's activations) stays within the
    int8 tolerance of the fp32 model: prediction maps relative L2 error <= 8e-2"""
    from oracle import net_int8_oracle as N
    case = TINY[1]
    tag, arch, size, classes = case[:4]
    layers, anchors, A, x = fp32_setup(case)
    r = F.detect(arch, layers, x, size, anchors, classes)
    mx = [np.abs(t).max() for t in r["taps"]] + [np.abs(p).max() for p in r["preds"]]
    sa = [O.floor_log2_scale(m)[0] for m in mx]
    ri = N.tiny_detect(x, N.quantize_folded(N.fold_bn(layers)), O.floor_log2_scale(np.abs(x).max())[0], sa,
                       size, anchors, classes)
    for p, pf in zip(ri["preds"], r["preds"]):
        rel = np.sqrt(((p.astype(np.float64) - pf) ** 2).sum() / (pf.astype(np.float64) ** 2).sum())
        assert rel <= 8e-2


@pytest.mark.gpu
@pytest.mark.parametrize("case", TINY, ids=[c[0] for c in TINY])
def test_int8_tiny_bit_exact(case, gold):
    """int8 engine == integer oracle on every tensor (bit-exact), detections equal up to score ties;
    exponents come from the product's own calibration (bf16 run); and the int8 maps stay within 8e-2
    relative L2 of the reference's fp32 maps"""
    from oracle import net_int8_oracle as N
    from yolo355 import prep
    from yolo355.netengine import Net
    tag, arch, size, classes = case[:4]
    fnet, layers, anchors, x = _load_net(case, len(case[5]))
    B = x.shape[0]
    sa_in, sa = fnet.calibration_exponents(x)
    # the calibration rule is the tracker's: exponents equal those of the fp32 oracle's maxima
    # up to one step (bf16 rounding can move a maximum across a power of two)
    r32 = F.detect(arch, layers, x, size, anchors, classes)
    mx = [np.abs(t).max() for t in r32["taps"]] + [np.abs(p).max() for p in r32["preds"]]
    assert all(abs(a - O.floor_log2_scale(m)[0]) <= 1 for a, m in zip(sa, mx))
    fnet.close()
    folded = N.fold_bn(layers)
    qprod = prep.quantize_folded(folded)
    qor = N.quantize_folded(folded)
    for a, b in zip(qprod, qor):
        assert a["e_w"] == b["e_w"] and a["e_b"] == b["e_b"]
        assert np.array_equal(a["q_w"], b["q_w"]) and np.array_equal(a["q_b"], b["q_b"])
    net = Net(arch, size, classes, anchors, 0.01, 0.5, max_batch=B, device="cuda:0", dtype="int8")
    for i, q in enumerate(qprod):
        net.load_layer_i8(i, q["q_w"], q["q_b"], q["e_w"], q["e_b"])
    net.set_act_exponents(sa_in, sa)
    sa_in2, sa_eff = net.get_act_exponents()
    out = net.forward(x, tap=True)
    ref = N.tiny_detect(x, qor, sa_in, sa, size, anchors, classes)
    assert sa_eff == ref["sa"] and sa_in2 == sa_in
    for t in range(net.num_tensors):
        got = np.rint(net.get_tensor(t, B).astype(np.float64) * 2.0 ** sa_eff[t]).astype(np.int64)
        assert np.array_equal(got, ref["t"][t]), "tensor %d differs in %d places" % (t, int((got != ref["t"][t]).sum()))
    assert net.counters() == ref["sat"]
    cb, cs, cc = net.candidates(B)
    assert np.allclose(cb, ref["box"], atol=2e-5, rtol=0)
    assert np.allclose(cs, ref["cls_scores"].max(axis=2), atol=2e-6, rtol=1e-5)
    for bi in range(B):
        ok, msg = dets_match(ref["dets"][bi][:3], out[bi], all_scores=cs[bi])
        assert ok, "%s image %d: %s" % (tag, bi, msg)
    for k, n in enumerate(("pred_1", "pred_2")):
        g = gold["%s/%s" % (tag, n)].astype(np.float64)
        rel = np.sqrt(((ref["preds"][k] - g) ** 2).sum() / (g ** 2).sum())
        assert rel <= 8e-2, "%s %s: int8 vs fp32 reference, relative L2 error %.3g" % (tag, n, rel)
    # round 4: a forward without the tap runs conv_1 + pool and conv_2 + pool in ONE launch (front.hip, the q_bf engine's fused
    # front end with the 205 / 2048 slope): every tensor from conv_2's on, the counters and the detections are the same
    # (conv_1's own map is not written by that launch: it still holds the tap forward's)
    tap_t = [net.get_tensor(t, B) for t in range(net.num_tensors)]
    out2 = net.forward(x)
    from yolo355 import _ffi
    with pytest.raises(_ffi.Y355Error) as ei:              # conv_1's map was not written by this forward
        net.get_tensor(0, B)
    assert ei.value.code == _ffi.ENOTREADY
    for t in range(1, net.num_tensors):
        assert np.array_equal(net.get_tensor(t, B), tap_t[t]), "tensor %d differs between the fused and the layer-by-layer front end" % t
    assert net.counters() == ref["sat"]
    for bi in range(B):
        assert all(np.array_equal(a, b) for a, b in zip(out[bi], out2[bi]))
    # Y355_NET_OPT_WORKGROUPS: fewer persistent workgroups per convr launch + one pair-walk workgroup per image
    # change the schedule only -- every tensor, the counters and the detection lists are the same
    for n in (7, 128, 0):
        net.set_option(_ffi.NET_OPT_WORKGROUPS, n)
        out3 = net.forward(x)
        for t in range(1, net.num_tensors):
            assert np.array_equal(net.get_tensor(t, B), tap_t[t]), "workgroups %d: tensor %d differs" % (n, t)
        assert net.counters() == ref["sat"]
        for bi in range(B):
            assert all(np.array_equal(a, b) for a, b in zip(out[bi], out3[bi])), "workgroups %d: image %d" % (n, bi)
    net.close()


@pytest.mark.gpu
def test_tiny_dropin_quantized():
    """YOLOv3tiny(x, quantization=True): first call freezes the exponents, later calls reuse them"""
    from yolo355.models import YOLOv3tiny
    case = TINY[1]
    tag, arch, size, classes = case[:4]
    sd, layers, anchors, x = _state_dict_of(case)
    m = YOLOv3tiny("cuda:0", input_size=size, num_classes=classes, anchor_size=anchors)
    m.load_state_dict(sd, strict=False)
    m.eval()
    xt = torch.from_numpy(x)
    q0 = m.forward_batch(xt, quantization=True)
    frozen = m.act_exponents
    q1 = m.forward_batch(xt[1:2], quantization=True)
    assert m.act_exponents is frozen
    assert np.array_equal(q0[1][0], q1[0][0]) and np.array_equal(q0[1][1], q1[0][1])
    f0 = m.forward_batch(xt)
    for bi in range(x.shape[0]):
        # int8 maps differ from the bf16 ones by ~4 % (test_int8_tiny_bit_exact), and greedy NMS
        # amplifies that: a sanity bound only
        fr, fg = dets_close(f0[bi], q0[bi], 0.5, 0.2)
        assert fr >= 0.6 and fg >= 0.6
        assert abs(len(f0[bi][1]) - len(q0[bi][1])) <= 0.15 * len(f0[bi][1])


@pytest.mark.gpu
@pytest.mark.parametrize("case", FP32_CASES, ids=IDS)
def test_bf16_fused_front_matches_layer_launches(case):
    """round 4: a forward without the tap runs conv(3->16)+pool and conv(16->32)+pool of the bf16 nets in ONE launch
    (csrc/frontb.hip); the tap forward runs them as two.  Same operands, same rounding points, only the order of the fp32
    accumulation differs: conv2's pooled map differs in 0.005-0.007 % of its values (one bf16 ulp; rel. L2 2-4e-5, measured), and the
    few flipped ulps reach the prediction maps as 1.4e-3 .. 2.1e-3 rel. L2 (measured; the bf16 path's distance to the fp32
    reference is 3e-3 .. 7e-3): bounds 1e-3 / 5e-3."""
    fnet, layers, anchors, x = _load_net(case, len(case[5]))
    B = x.shape[0]
    fused = fnet.forward(x)
    nt = fnet.num_tensors
    npred = 2 if case[1] == "tiny_yolo_v3" else 1
    # the fused launch does not write conv1's own map: reading it (or its maximum) is an error, not stale data
    from yolo355 import _ffi
    for read in (lambda: fnet.get_tensor(0, B), lambda: fnet.tensor_absmax(0, B)):
        with pytest.raises(_ffi.Y355Error) as ei:
            read()
        assert ei.value.code == _ffi.ENOTREADY
    t1_f = fnet.get_tensor(1, B).astype(np.float64)
    pred_f = [fnet.get_tensor(nt - npred + k, B).astype(np.float64) for k in range(npred)]
    tap = fnet.forward(x, tap=True)
    t1_t = fnet.get_tensor(1, B).astype(np.float64)
    pred_t = [fnet.get_tensor(nt - npred + k, B).astype(np.float64) for k in range(npred)]
    rel1 = np.sqrt(((t1_f - t1_t) ** 2).sum() / (t1_t ** 2).sum())
    differing = float((t1_f != t1_t).mean())
    print("conv2 map: rel. L2 %.2e, %.3f %% of the values differ; pred rel. L2 %s" % (
        rel1, 100 * differing, ["%.2e" % np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()) for a, b in zip(pred_f, pred_t)]))
    assert rel1 <= 1e-3 and differing <= 0.05
    assert np.abs(t1_f - t1_t).max() <= 2 ** -6 * max(1.0, np.abs(t1_t).max())
    for a, b in zip(pred_f, pred_t):
        assert np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()) <= 5e-3
    for bi in range(B):
        fr, fg = dets_close(tap[bi], fused[bi], 0.9, 0.02)
        assert fr >= 0.9 and fg >= 0.9
    fnet.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", FP32_CASES[:3], ids=IDS[:3])
def test_convpxb_route_against_the_ring_route(case):
    """Round 6: SlimYOLOv2's thin 3x3 layers (conv3_1, conv3_2 + pool3, conv4_1) with the weights in registers (csrc/convpxb.hip, the
    default) against the LDS-ring kernels they replace (Y355_NET_OPT_THIN_RESIDENT = 0): the same bf16 operands and fp32
    accumulation, the bias entering the sum first instead of last -- every element within 2 bf16 ulps, 99.9 % identical or 1 ulp
    apart; sizes: 416 x 416, a non-square map with edge groups (320 x 416), maps narrower than two DMA pieces (96 x 160), B = 1 / 2."""
    from yolo355 import _ffi
    tag, arch, size, classes = case[:4]
    net, layers, anchors, x = _load_net(case, len(case[5]))
    B = x.shape[0]
    net.forward(x, tap=True)
    new = [net.get_tensor(k, B).copy() for k in range(net.num_tensors)]
    net.set_option(_ffi.NET_OPT_THIN_RESIDENT, 0)
    net.forward(x, tap=True)
    old = [net.get_tensor(k, B).copy() for k in range(net.num_tensors)]
    # tensors of the graph: 0 conv1 + pool, 1 conv2 + pool, 2 conv3_1, 3 conv3_2 + pool, 4 conv4_1, 5 conv4_2 + pool, 6..8 conv5..7, 9 pred
    def ulps(a, b):                                       # bf16 values held in fp32: one ulp of max(|a|, |b|) is 2^(exponent - 7)
        m = np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-30)
        return np.abs(a.astype(np.float64) - b) / np.ldexp(1.0, np.floor(np.log2(m)).astype(np.int64) - 7)
    for k in (0, 1):
        assert np.array_equal(new[k], old[k]), k          # in front of the thin layers: the same launches
    d = ulps(new[2], old[2])                              # conv3_1: identical inputs on both routes
    # within 2 bf16 ulps of the value -- or, where the sum cancels to almost nothing, within 2^-10 of the map's r.m.s. (the fp32
    # accumulation orders differ by ~1e-7 of the terms' magnitude, which is many "ulps" of a result near zero)
    rms = float(np.sqrt((old[2].astype(np.float64) ** 2).mean()))
    small = np.abs(new[2].astype(np.float64) - old[2]) <= rms * 2.0 ** -10
    assert ((d <= 2.0) | small).all() and (d <= 1.0).mean() >= 0.999, (float(d[~small].max()) if (~small).any() else 0.0, float((d <= 1.0).mean()))
    assert (d > 0).any(), "the option did not change the route"
    for k in range(3, len(new)):                          # behind it the one-ulp differences propagate: relative L2, and most values untouched
        rel = np.sqrt(((new[k].astype(np.float64) - old[k]) ** 2).sum() / max((old[k].astype(np.float64) ** 2).sum(), 1e-30))
        assert rel <= (3e-3 if k < len(new) - 1 else 5e-3), (k, rel)
        if k in (3, 4):
            assert (ulps(new[k], old[k]) <= 1.0).mean() >= 0.99, k
    with pytest.raises(_ffi.Y355Error):
        net.set_option(_ffi.NET_OPT_THIN_RESIDENT, 2)
    net.close()
