"""Round-2 parity tests against tests/golden/r2.npz (gen_golden_r2.py: outputs of the reference itself).

CPU: the helpers and the oracle against the goldens (BaseTransform, ReLU epilogue, EMA trackers).
GPU: the default `net(x)` call (quantization=False) of the q_bf drop-in, the ReLU epilogue through the C ABI, multi-batch
EMA calibration, stream switching, and full-batch runs of configs 3 and 4 (B = 64 / 128).
"""
import os

import numpy as np
import pytest
import torch

from cases import FP32_CASES, fp32_setup
from helpers import crc, dets_close, dets_match
from oracle import yolo_oracle as O
from yolo355 import prep, synth

TRACKERS = ["a_tracker_in", "a_tracker1", "a_tracker2", "a_tracker3_1", "a_tracker3_2", "a_tracker4_1",
            "a_tracker4_2", "a_tracker5", "a_tracker6", "a_tracker7", "a_tracker_pred"]
# must mirror tests/golden/gen_golden_r2.py
QF32_CASES = [
    ("qf_416", dict(seed=2, weight_gain=2.2, pred_gain=1.5, obj_bias=-2.0), [416, 416], 2, "mask", [0], "blocks"),
    ("qf_b2", dict(seed=5, weight_gain=2.2, pred_gain=1.5, obj_bias=-2.0), [240, 320], 2, "mask", [22, 23], "noise"),
    ("qf_voc", dict(seed=6, weight_gain=2.2, pred_gain=1.5, obj_bias=-2.0), [96, 160], 20, "voc", [7], "blocks"),
]
EMA = dict(weights=dict(seed=2), size=[96, 160], classes=2, batch=2, seeds=[41, 42, 43, 44, 45])


def _same_list(msg):
    """dets_match verdicts that mean "the same detections in the same order": identical lists, or identical boxes / classes
    with every score inside the tolerance (the strict pass also bounds every box coordinate by BOX_TOL; a coordinate a hair
    past it that still rounds to the same grid cell reports "same boxes").  What is NOT accepted here is the tie-group
    fallback: oracle and engine order ties identically by definition."""
    return msg in ("exact", "empty") or msg.startswith("same boxes")
@pytest.fixture(scope="module")
def r2():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "r2.npz"))


def _rand_i8(seed, shape):
    return (synth.uniform_u8(seed, shape).astype(np.int32) - 128).clip(-127, 127)


# ------------------------------------------------------------------------------------------ CPU
def test_normalize_frames_is_the_reference_base_transform(r2):
    """synth.normalize_frames == the reference's BaseTransform (data/__init__.py:30-56, resize = identity at the network
    size) + test.py:79-80, bit for bit.  With tests/test_gpu_parity.py::test_forward_frames_equals_normalised_tensor
    (y355_forward_u8 on the frames == y355_forward on normalize_frames(frames)) this pins the uint8 route to the reference."""
    n = 0
    while "bt/%d/meta" % n in r2:
        h, w, seed, pat = [int(v) for v in r2["bt/%d/meta" % n]]
        frames = synth.make_frames_u8(seed, 1, h, w, ["noise", "blocks"][pat])
        x = synth.normalize_frames(frames)
        assert x.dtype == np.float32 and x.shape == (1, 3, h, w)
        assert crc(x) == int(r2["bt/%d/crc" % n][0]), n
        if "bt/%d/x" % n in r2:
            assert np.array_equal(x, r2["bt/%d/x" % n])
        n += 1
    assert n == 3


def test_oracle_relu_matches_reference(r2):
    n = 0
    while "relu/%d/meta" % n in r2:
        cin, cout, h, w, sa_in, e_w, e_b, sa_out, nobias, s0, s1, s2 = [int(v) for v in r2["relu/%d/meta" % n]]
        q_in, q_w = _rand_i8(s0, (2, cin, h, w)), _rand_i8(s1, (cout, cin, 3, 3))
        q_b = np.zeros(cout, np.int32) if nobias else _rand_i8(s2, (cout,))
        t, Fx, _ = O.conv_layer_int(q_in, q_w, q_b, sa_in, e_w, 0 if nobias else e_b, "relu")
        assert np.array_equal(t.astype(np.float32) * np.float32(2.0 ** -Fx), r2["relu/%d/y" % n])
        assert np.array_equal(O.rne_shift(t, Fx - sa_out), r2["relu/%d/q_out" % n])
        n += 1
    assert n == 3


def _ema_batches():
    return [synth.make_images(s, EMA["batch"], EMA["size"][0], EMA["size"][1]) for s in EMA["seeds"]]


def test_oracle_ema_matches_reference(r2):
    """AveragedRangeTracker's non-frozen branch (models/slim_yolo_v2.py:30-31) over five batches, weights quantized
    first: the oracle's trackers reproduce the reference's scales and exponents after every batch."""
    ql = O.quantize_layers(synth.make_weights(**EMA["weights"], num_classes=EMA["classes"]))
    tr = [O.RangeTracker() for _ in range(11)]
    for it, x in enumerate(_ema_batches()):
        r = O.forward_backbone_int(x, ql, tr, quant_freeze=False, keep=False)
        assert r["sa"] == [int(v) for v in r2["ema/prequant/sa"][it]], it
        sc = np.array([float(t.scale.item()) for t in tr], np.float32)
        assert np.allclose(sc, r2["ema/prequant/scale"][it], rtol=1e-6, atol=0), it
        assert sum(r["sat"]) == 0            # no value left the int8 range: the reference (no clamp) and the engine agree


def test_ema_script_order_differs_only_in_the_first_batch(r2):
    """retune_bias_quantize.py:357-369 runs its FIRST calibration batch on the un-quantized weights (quantize_layers is
    called after the forward); the engine's loop quantizes first.  Recorded: how far the two orders drift apart."""
    a, b = r2["ema/script/scale"], r2["ema/prequant/scale"]
    assert np.array_equal(r2["ema/script/sa"][-1], r2["ema/prequant/sa"][-1])
    assert np.abs(a / b - 1).max() < 0.02


def test_dets_match_reports_its_mode():
    b = np.array([[0.1, 0.1, 0.3, 0.3], [0.5, 0.5, 0.9, 0.9]], np.float32)
    s = np.array([0.9, 0.8], np.float32)
    c = np.array([0, 1], np.int64)
    assert dets_match((b, s, c), (b, s, c)) == (True, "exact")
    ok, msg = dets_match((b, s, c), (b[:1], s[:1], c[:1]), all_scores=np.array([0.9, 0.8, 0.8]))
    assert ok and msg.startswith("tie-tolerant")
    ok, msg = dets_match((b, s, c), (b[:1], s[:1], c[:1]), all_scores=np.array([0.9, 0.8, 0.7]))
    assert not ok


# ------------------------------------------------------------------------------------------ GPU
def _dyadic_state(net, weights):
    """the checkpoint quantize_layers leaves behind: every conv tensor = q / 2^e (retune_bias_quantize.py:111-119)"""
    sd = net.state_dict()
    for name, w, b in weights:
        k = "pred" if name == "pred" else name + ".convs.0"
        for suffix, t in ((".weight", w), (".bias", b)):
            q, e = prep.to_int8_pow2(torch.from_numpy(t))
            sd[k + suffix] = torch.from_numpy(q.astype(np.float32) * np.float32(2.0 ** -e))
    net.load_state_dict(sd, strict=False)
    return net.eval()


@pytest.mark.gpu
@pytest.mark.parametrize("case", QF32_CASES, ids=[c[0] for c in QF32_CASES])
def test_default_call_quantization_false(case, r2):
    """`net(x)` exactly as test.py:84 / demo.py:81 / utils/vocapi_evaluator.py:67 call the q_bf model: quantization=False,
    trackers = identity, fp32 math on the loaded dyadic weights (models/slim_yolo_v2.py:212-358).  The drop-in runs it on
    the bf16 MFMA: tolerances of tests/test_fp32_models.py against the reference's own outputs."""
    from yolo355.models.slim_yolo_v2 import SlimYOLOv2_quantize_bnfuse
    tag, wkw, size, classes, an, seeds, pattern = case
    anchors = synth.ANCHOR_SIZE_MASK if an == "mask" else synth.ANCHOR_SIZE
    net = SlimYOLOv2_quantize_bnfuse("cuda:0", input_size=size, num_classes=classes, trainable=False, conf_thresh=0.01,
                                     nms_thresh=0.5, anchor_size=anchors)
    _dyadic_state(net, synth.make_weights(**wkw, num_classes=classes))
    x = np.concatenate([synth.make_images(s, 1, size[0], size[1], pattern) for s in seeds])
    xt = torch.from_numpy(x)
    out = net(xt)                                                     # the canonical call
    assert [type(o) for o in out] == [np.ndarray] * 3 and out[0].dtype == np.float32 and out[2].dtype == np.int64
    assert out[0].flags.writeable
    allimg = net.forward_batch(xt, quantization=False)
    assert all(np.array_equal(a, b) for a, b in zip(out, allimg[0]))
    f32 = net._get_f32_net(len(seeds), False)
    got = f32.get_tensor(f32.num_tensors - 1, len(seeds)).astype(np.float64)
    g = r2[tag + "/pred"].astype(np.float64)
    rel = np.sqrt(((got - g) ** 2).sum() / (g ** 2).sum())
    assert rel <= 1.5e-2, "%s: pred relative L2 error %.3g" % (tag, rel)
    assert np.abs(got - g).max() <= 2.5e-2 * np.abs(g).max()
    for conf in (0.01, 0.1):
        net.conf_thresh = conf
        dets = net.forward_batch(xt, quantization=False)
        for bi in range(len(seeds)):
            ref = tuple(r2["%s/%d/det%g/%s" % (tag, bi, conf, k)] for k in ("boxes", "scores", "cls"))
            fr, fg = dets_close(ref, dets[bi], 0.8, 0.05)
            assert fr >= 0.8 and fg >= 0.8, "%s image %d conf %g: matched %.3f / %.3f" % (tag, bi, conf, fr, fg)
            fr, fg = dets_close(ref, dets[bi], 0.5, 0.2)
            assert fr >= 0.9 and fg >= 0.9
            assert abs(len(dets[bi][1]) - len(ref[1])) <= max(0.03 * len(ref[1]), 5)
    # the evaluators' in-place rescale (utils/vocapi_evaluator.py:69-70) through sizes_wh
    scaled = net.forward_batch(xt, quantization=False, sizes_wh=[[640, 480]] * len(seeds))
    assert np.allclose(scaled[0][0], dets[0][0] * np.array([[640, 480, 640, 480]], np.float32))


@pytest.mark.gpu
def test_relu_epilogue_matches_reference(r2):
    """Conv2d_fuse / Conv2d_fuse_nobias with leakyReLU=False (utils/modules.py:26,37): the C ABI's ReLU epilogue and the
    drop-in modules against the reference module's own output and its tracker-quantized value."""
    from yolo355.engine import conv3x3_i8_fused
    from yolo355.utils import Conv2d_fuse, Conv2d_fuse_nobias
    n = 0
    while "relu/%d/meta" % n in r2:
        cin, cout, h, w, sa_in, e_w, e_b, sa_out, nobias, s0, s1, s2 = [int(v) for v in r2["relu/%d/meta" % n]]
        q_in, q_w = _rand_i8(s0, (2, cin, h, w)), _rand_i8(s1, (cout, cin, 3, 3))
        q_b = np.zeros(cout, np.int32) if nobias else _rand_i8(s2, (cout,))
        ref_q = r2["relu/%d/q_out" % n]
        out, st = conv3x3_i8_fused(q_in, q_w, q_b, sa_in, e_w, 0 if nobias else e_b, sa_out, leaky=False, relu=True)
        assert np.array_equal(out.astype(np.int32), np.clip(ref_q, -127, 127)), n
        assert st["saturated"] == int((np.abs(ref_q) > 127).sum())
        m = (Conv2d_fuse_nobias if nobias else Conv2d_fuse)(cin, cout, 3, 1, leakyReLU=False)
        with torch.no_grad():
            m.convs[0].weight.copy_(torch.from_numpy(q_w.astype(np.float32) / np.float32(2.0 ** e_w)))
            if not nobias:
                m.convs[0].bias.copy_(torch.from_numpy(q_b.astype(np.float32) / np.float32(2.0 ** e_b)))
        y = m(torch.from_numpy(q_in.astype(np.float32) / np.float32(2.0 ** sa_in)).cuda())
        assert np.array_equal(y.cpu().numpy(), r2["relu/%d/y" % n]), n
        n += 1
    assert n == 3


@pytest.mark.gpu
def test_ema_calibration_matches_reference(r2):
    """Engine.calibrate(freeze=False) over five batches = the reference's trackers in training mode (EMA, :30-31), weights
    quantized before the first batch: identical scales and exponents after every batch."""
    from yolo355.engine import Engine
    ql = O.quantize_layers(synth.make_weights(**EMA["weights"], num_classes=EMA["classes"]))
    eng = Engine(EMA["size"], EMA["classes"], synth.ANCHOR_SIZE_MASK, max_batch=EMA["batch"])
    eng.load_quantized(ql)
    tr = [prep.RangeTracker() for _ in range(11)]
    for it, x in enumerate(_ema_batches()):
        sa = eng.calibrate(x, tr, freeze=False)
        assert sa == [int(v) for v in r2["ema/prequant/sa"][it]], it
        sc = np.array([float(t.scale.item()) for t in tr], np.float32)
        assert np.allclose(sc, r2["ema/prequant/scale"][it], rtol=1e-6, atol=0), it
    eng.close()


@pytest.mark.gpu
def test_engine_follows_the_callers_stream():
    """An engine built under one stream and called under another: the forward must see the caller's pending
    writes and the caller must see the forward's results without an explicit synchronize."""
    from yolo355.engine import Engine
    dev = torch.device("cuda", 0)
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2, pred_gain=400.0, obj_bias=-4.0))
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s1):
        eng = Engine([96, 160], 2, synth.ANCHOR_SIZE_MASK, max_batch=2, device=dev)
        eng.load_quantized(ql)
    xh = synth.make_images(3, 2, 96, 160, "blocks")
    eng.calibrate(xh[:1], [prep.RangeTracker() for _ in range(11)])
    ref = eng.forward(xh)
    assert sum(len(d[1]) for d in ref) > 0
    torch.cuda.synchronize()
    for it in range(20):
        with torch.cuda.stream(s2):
            xd = torch.zeros((2, 3, 96, 160), device=dev)
            big = torch.randn(4096, 4096, device=dev)
            for _ in range(3):
                big = big @ big * 1e-4                 # keeps s2 busy: the copy below is still pending at the forward's launch
            xd.copy_(torch.from_numpy(xh), non_blocking=True)
            ob, os_, oc, on = eng.forward_device(xd)
            n = on[:2].cpu().numpy()                   # read on s2, right behind the forward on s1
            sc = os_[:2].cpu().numpy()
        for i in range(2):
            assert n[i] == len(ref[i][1]), (it, i)
            assert np.array_equal(sc[i, :n[i]], ref[i][1])
    eng.close()


# ---- configs 3 and 4 at their full batch (only B = 1-2 was exercised on the GPU)
def _full_batch_net(case, B, dtype):
    from yolo355.netengine import Net
    from oracle import fp32_oracle as F
    tag, arch, size, classes = case[:4]
    layers, anchors, A, x = fp32_setup(case)
    folded = []
    for L in layers:
        w, b = L["w"].astype(np.float64), L["b"].astype(np.float64)
        if L["bn"] is not None:
            g, be, mu, var = (a.astype(np.float64) for a in L["bn"])
            s = g / np.sqrt(var + F.EPS)
            w, b = w * s[:, None, None, None], (b - mu) * s + be
        folded.append((w.astype(np.float32), b.astype(np.float32)))
    fnet = Net(arch, size, classes, anchors, 0.01, 0.5, max_batch=B, device="cuda:0", dtype="bf16")
    for i, (w, b) in enumerate(folded):
        fnet.load_layer(i, w, b)
    if dtype == "bf16":
        return fnet, None, layers, anchors, folded
    sa_in, sa = fnet.calibration_exponents(x[:1])
    qnet = Net(arch, size, classes, anchors, 0.01, 0.5, max_batch=B, device="cuda:0", dtype="int8")
    for i, q in enumerate(prep.quantize_folded(folded)):
        qnet.load_layer_i8(i, q["q_w"], q["q_b"], q["e_w"], q["e_b"])
    qnet.set_act_exponents(sa_in, sa)
    fnet.close()
    return qnet, (sa_in, sa), layers, anchors, folded


def _periodic_and_independent(net, base, B, k):
    """size-independent properties of a full batch: images repeat with period k, so detections and prediction maps must
    repeat; a sub-batch equals the same elements of the full batch (batch independence)"""
    x = np.concatenate([base] * (B // k))
    dets = net.forward(x)
    pred = [net.get_tensor(t, B) for t in range(net.num_tensors - (2 if net.arch == "tiny_yolo_v3" else 1), net.num_tensors)]
    for i in range(k, B):
        for p in pred:
            assert np.array_equal(p[i], p[i % k]), i
        for a, b in zip(dets[i], dets[i % k]):
            assert np.array_equal(a, b), i
    sub = net.forward(x[:k + 3])
    for i in range(k + 3):
        for a, b in zip(sub[i], dets[i]):
            assert np.array_equal(a, b), i
    return dets, pred


def _attribute_list_differences(net, x1, ref_box, ref_prob, conf=0.01, nms=0.5, score_tol=0.04, box_tol=0.08):
    """every difference between the reference's detection list and the engine's must be ATTRIBUTED -- to a
    per-anchor deviation inside the stated per-anchor tolerances (a score that crosses conf_thresh, a best class that flips,
    a pair's IoU that crosses nms_thresh) or to a decision downstream of one (cascade) -- by replaying both sides' greedy NMS
    on the per-anchor candidates (helpers.explain_detection_differences).  Unattributed remainder: 0."""
    from helpers import explain_detection_differences
    got = net.forward(x1, tap=True)[0]
    cb, cs, cc = net.candidates(1)
    r = explain_detection_differences(ref_box, ref_prob.max(axis=1), ref_prob.argmax(axis=1), cb[0], cs[0], cc[0], conf, nms)
    assert r["n_got"] == len(got[1]), "the replayed NMS is not the engine's: %d vs %d detections" % (r["n_got"], len(got[1]))
    # the per-anchor tolerances themselves (tests/test_fp32_models.py holds them on every anchor)
    assert np.abs(cs[0] - ref_prob.max(axis=1)).max() <= score_tol
    assert r["unexplained"] == [], "%d detection-list differences have no per-anchor cause: anchors %s" % (len(r["unexplained"]), r["unexplained"][:10])
    assert r["root_score_dev"] <= score_tol and r["root_box_dev"] <= box_tol, r
    return r


@pytest.mark.gpu
def test_config3_full_batch_slim_fp32(r2):
    """BASELINE configs[2]: SlimYOLOv2 (fp32 weights, bf16 MFMA), B = 64, 416 x 416: first image against the reference's fp32
    golden (tests/golden/fp32.npz) at the stated bf16 tolerances + periodicity + batch independence."""
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "fp32.npz"))
    case = FP32_CASES[0]                                    # slim416
    tag, arch, size, classes = case[:4]
    layers, anchors, A, x0 = fp32_setup(case)
    net, _, _, _, _ = _full_batch_net(case, 64, "bf16")
    base = np.concatenate([x0, synth.make_images(101, 3, size[0], size[1])])
    dets, pred = _periodic_and_independent(net, base, 64, 4)
    g = gold[tag + "/pred"].astype(np.float64)
    rel = np.sqrt(((pred[0][:1] - g) ** 2).sum() / (g ** 2).sum())
    assert rel <= 1.5e-2, rel
    ref = tuple(gold["%s/0/det0.01/%s" % (tag, k)] for k in ("boxes", "scores", "cls"))
    fr, fg = dets_close(ref, dets[0], 0.8, 0.05)
    print("config 3, image 0: pred rel. L2 %.4f; %.3f of the reference's and %.3f of the engine's detections matched "
          "(same class, IoU >= 0.8, |score err| <= 0.05); %d vs %d detections" % (rel, fr, fg, len(ref[1]), len(dets[0][1])))
    # measured on the MI355X (round 4, printed above with -s): rel. L2 0.0068, 0.840 / 0.841 matched, 2333 vs 2329 detections;
    # the assertions sit a few points under the measurement
    assert fr >= 0.82 and fg >= 0.82 and abs(len(ref[1]) - len(dets[0][1])) <= 0.02 * len(ref[1])
    # ... and what the unmatched ~16 % are: the reference's own prediction map decoded per anchor (the reference's head, restated
    # in the oracle) against the engine's per-anchor decode, both sides' NMS replayed
    rbox, rprob = O.head_decode(gold[tag + "/pred"], size, anchors, classes)
    r = _attribute_list_differences(net, x0, rbox[0], rprob[0])
    assert abs(r["n_ref"] - len(ref[1])) <= 2, "the replayed NMS is not the reference's: %d vs %d" % (r["n_ref"], len(ref[1]))
    print("config 3, image 0: %d of %d anchors end differently in the two lists: %s, unexplained 0; root deviations: score %.4f, box %.4f"
          % (r["differ"], len(rbox[0]), r["causes"], r["root_score_dev"], r["root_box_dev"]))
    net.close()


@pytest.mark.gpu
def test_config4_full_batch_tiny_int8():
    """BASELINE configs[3]: YOLOv3tiny int8, B = 128, 416 x 416: two images bit-exact against the integer oracle on every
    prediction map, detections tie-tolerantly equal, + periodicity + batch independence.  Against the reference's fp32
    model the int8 form is held to a stated DETECTION-level tolerance (the reference has no int8 tiny model: parity
    unpinned).  Measured on the MI355X (round 3, printed below with -s): 90.7 % of the reference's and 92.3 %
    of the engine's detections matched at (same class, IoU >= 0.5, |score err| <= 0.2), 1213 detections against 1230; the
    assertions sit a few points under the measurement (a layer with a wrong exponent or slope costs tens of points): >= 85 % /
    87 % matched, counts within 5 %."""
    from oracle import net_int8_oracle as N
    from oracle import fp32_oracle as F
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "fp32.npz"))
    case = [c for c in FP32_CASES if c[0] == "tiny416"][0]
    tag, arch, size, classes = case[:4]
    layers, anchors, A, x0 = fp32_setup(case)
    net, (sa_in, sa), _, _, folded = _full_batch_net(case, 128, "int8")
    base = np.concatenate([x0, synth.make_images(201, 1, size[0], size[1], "blocks")])
    dets, pred = _periodic_and_independent(net, base, 128, 2)
    sa_in2, sa_eff = net.get_act_exponents()
    ref = N.tiny_detect(base, N.quantize_folded(N.fold_bn(layers)), sa_in, sa, size, anchors, classes)
    nt = net.num_tensors
    for k in range(2):
        got = np.rint(pred[k][:2].astype(np.float64) * 2.0 ** sa_eff[nt - 2 + k]).astype(np.int64)
        assert np.array_equal(got, ref["t"][nt - 2 + k]), k
    for i in range(2):
        ok, msg = dets_match(ref["dets"][i][:3], dets[i], all_scores=ref["cls_scores"][i].max(axis=1))
        assert ok and _same_list(msg), (i, msg)         # integer oracle vs engine: same tie order by definition
    gref = tuple(gold["%s/0/det0.01/%s" % (tag, k)] for k in ("boxes", "scores", "cls"))
    fr, fg = dets_close(gref, dets[0], 0.5, 0.2)
    print("MEASURED config4 vs fp32 reference: matched fr=%.4f fg=%.4f, detections %d vs %d" % (fr, fg, len(dets[0][1]), len(gref[1])))
    assert fr >= 0.85 and fg >= 0.87, "int8 tiny vs the reference's fp32 detections: matched %.3f / %.3f" % (fr, fg)
    assert abs(len(dets[0][1]) - len(gref[1])) <= 0.05 * len(gref[1])
    # the unmatched detections attributed (see _attribute_list_differences); the int8 form's per-anchor tolerance against the
    # fp32 reference is wider than the bf16 form's: scores 0.15, boxes 0.25 (stated here, measured below)
    rbox, rprob = F.tiny_head_decode([torch.from_numpy(gold[tag + "/pred_1"]), torch.from_numpy(gold[tag + "/pred_2"])], size, anchors, classes)
    r = _attribute_list_differences(net, x0, np.asarray(rbox)[0], np.asarray(rprob)[0], score_tol=0.15, box_tol=0.25)
    assert abs(r["n_ref"] - len(gref[1])) <= 2, "the replayed NMS is not the reference's: %d vs %d" % (r["n_ref"], len(gref[1]))
    print("config 4, image 0: %d anchors end differently in the two lists: %s, unexplained 0; root deviations: score %.4f, box %.4f"
          % (r["differ"], r["causes"], r["root_score_dev"], r["root_box_dev"]))
    net.close()


@pytest.mark.gpu
def test_gather_of_real_engine_output_world1():
    """SURVEY.md 8e check "gathered detections equal the single-GPU run bit for bit" with what one GPU allows: the real
    engine's padded outputs through (a) shard.allgather_detections under the nccl (= RCCL) backend with world size 1 and
    (b) the C ABI route (y355_comm_init / y355_pack_dets / y355_allgather_dets / y355_unpack_dets); ragged case: the
    global batch is smaller than the engine's padded batch."""
    import socket
    import torch.distributed as dist
    from yolo355 import shard
    from yolo355.engine import Engine
    dev = torch.device("cuda", 0)
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2, pred_gain=400.0, obj_bias=-4.0))
    eng = Engine([96, 160], 2, synth.ANCHOR_SIZE_MASK, conf_thresh=0.05, max_batch=3, device=dev)
    eng.load_quantized(ql)
    x = synth.make_images(3, 3, 96, 160, "blocks")
    eng.calibrate(x[:1], [prep.RangeTracker() for _ in range(11)])
    direct = eng.forward(x)
    ob, os_, oc, on = eng.forward_device(torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    assert int(on[:3].sum()) > 0

    def check(g):
        got = shard.unpack(*g)
        assert len(got) == 3
        for a, b in zip(direct, got):
            for u, v in zip(a, b):
                assert np.array_equal(u, v)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        check(shard.allgather_detections(ob[:3], os_[:3], oc[:3], on[:3]))
        finish, works = shard.allgather_detections(ob[:3], os_[:3], oc[:3], on[:3], async_op=True)
        for w in works:
            w.wait()
        check(finish())
        rg = shard.RcclGather(1, 0, dev)
        g = rg.allgather(ob[:3], os_[:3], oc[:3], on[:3])
        torch.cuda.synchronize()
        check(g)
        # the C pack kernel and the torch pack produce the same bytes
        rec_t = shard.pack_detections(ob[:3], os_[:3], oc[:3], on[:3], records=4)
        assert torch.equal(rg._keep[0], rec_t[:3])
        rg.close()
    finally:
        dist.destroy_process_group()
    eng.close()


# ---- the resize stage of BaseTransform (cv2.resize, data/__init__.py:36): parity UNPINNED (OpenCV is not vendored by the
# reference and not installed here); the GPU stage is held bit-exact to the numpy restatement of OpenCV's 8-bit algorithm
def test_resize_oracle_properties():
    from oracle.resize_oracle import resize_linear_u8, linear_tables
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (2, 37, 53, 3), dtype=np.uint8)
    assert np.array_equal(resize_linear_u8(a, 37, 53), a)                       # equal sizes: a copy
    b = rng.integers(0, 256, (1, 40, 60, 3), dtype=np.uint8)
    area = ((b[:, 0::2, 0::2].astype(int) + b[:, 0::2, 1::2] + b[:, 1::2, 0::2] + b[:, 1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert np.array_equal(resize_linear_u8(b, 20, 30), area)                    # exact 2x decimation = OpenCV's area-fast path
    assert np.unique(resize_linear_u8(np.full((1, 11, 17, 3), 200, np.uint8), 416, 416)).tolist() == [200]
    ofs, coef = linear_tables(4, 8)                                             # 2x upscale: quarter-pixel phases, clamped borders
    assert ofs.tolist() == [0, 0, 0, 1, 1, 2, 2, 3]
    assert coef.tolist() == [[2048, 0], [1536, 512], [512, 1536], [1536, 512], [512, 1536], [1536, 512], [512, 1536], [2048, 0]]
    up = resize_linear_u8(np.array([[[[0, 0, 0], [100, 100, 100]]]], np.uint8), 1, 4)[0, 0, :, 0]
    assert up.tolist() == [0, 25, 75, 100]
    # vertical axis: OpenCV keeps floor(f) and the coefficient pair and clips the ROW INDICES, so a border row is
    # blended with itself through two truncated products: 255 -> ((512 * 32640) >> 16) + ((1536 * 32640) >> 16) + 2 >> 2 = 255,
    # but 77 -> (77 + 230 + 2) >> 2 = 77 while 3 -> (3 + 8 + 2) >> 2 = 3, 5 -> (4 + 14 + 2) >> 2 = 5, 6 -> (5 + 17 + 2) >> 2 = 6
    ofs, coef = linear_tables(4, 8, vertical=True)
    assert ofs.tolist() == [-1, 0, 0, 1, 1, 2, 2, 3] and coef[0].tolist() == [512, 1536] and coef[7].tolist() == [1536, 512]
    col = rng.integers(0, 256, (1, 5, 1, 3), dtype=np.uint8)
    upv = resize_linear_u8(col, 10, 1)
    v = col[0, 0, 0].astype(int) * 2048 >> 4
    assert upv[0, 0, 0].tolist() == ((((512 * v) >> 16) + ((1536 * v) >> 16) + 2) >> 2).tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", [((480, 640), (416, 416)), ((375, 500), (320, 416)), ((120, 160), (240, 320)),
                                     ((96, 160), (96, 160)), ((833, 417), (416, 416))])
def test_resize_stage_matches_oracle(src, dst):
    """y355_forward_u8_resized: the resized frames equal the oracle's bit for bit (down- and up-scaling, non-integer ratios,
    odd sizes, equal sizes), and the detections equal y355_forward_u8 on the oracle-resized frames."""
    from oracle.resize_oracle import resize_linear_u8
    from yolo355.engine import Engine
    B = 2
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2, pred_gain=400.0, obj_bias=-4.0))
    eng = Engine(list(dst), 2, synth.ANCHOR_SIZE_MASK, conf_thresh=0.05, max_batch=B)
    eng.load_quantized(ql)
    frames = synth.make_frames_u8(77, B, src[0], src[1], "blocks")
    want = resize_linear_u8(frames, dst[0], dst[1])
    got = eng.resize_frames(frames).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want), int((got != want).sum())
    eng.calibrate(synth.normalize_frames(want[:1]), [prep.RangeTracker() for _ in range(11)])
    a = eng.forward_frames(frames)                      # any size in: resize on the GPU, then the fused uint8 route
    b = eng.forward_frames(want)
    assert sum(len(d[1]) for d in a) > 0
    for u, v in zip(a, b):
        for s, t in zip(u, v):
            assert np.array_equal(s, t)
    eng.close()


# ---------------------------------------------------------------- NMS kernels in isolation (pairs + rounds), round-2 rewrite
@pytest.mark.gpu
@pytest.mark.parametrize("nms_thr", [0.5, 0.3, 0.75, 5e-5])
def test_nms_sweep_on_the_engines_own_candidates(nms_thr):
    """The round-2 pair walk (two workgroups per image, 64-lane runs in passes, L lanes per candidate on small images,
    branch-free test, per-wave edge buffers) and the register-resident rounds, against the oracle's greedy NMS
    (models/slim_yolo_v2.py:145-210, tie order (score desc, anchor index asc)) run on the SAME decoded candidates: exact
    equality.  Densities: every anchor a candidate (3380: two passes of runs), about a third, a handful (L = 8);
    nms_thresh 5e-5 is outside the range the pruning bounds hold for (the un-pruned kernel instantiation)."""
    from yolo355.engine import Engine
    B, C = 3, 2
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=C))
    eng = Engine([416, 416], C, synth.ANCHOR_SIZE_MASK, conf_thresh=0.01, nms_thresh=nms_thr, max_batch=B)
    eng.load_quantized(ql)
    eng.calibrate(synth.make_images(1, 1, 416, 416), [prep.RangeTracker() for _ in range(11)])
    x = synth.make_images(1000, B, 416, 416)
    seen = []
    for conf in (0.01, 0.45, 0.62):
        eng.set_thresholds(conf, nms_thr)
        dets = eng.forward(x, tap=True)
        cb, cs, cc = eng.candidates(B)
        for i in range(B):
            prob = np.zeros((cb.shape[1], C), np.float32)
            prob[np.arange(cb.shape[1]), cc[i]] = cs[i]
            ref = O.postprocess(cb[i], prob, conf, nms_thr, C)[:3]
            seen.append(int((cs[i] >= np.float32(conf)).sum()))
            assert len(ref[1]) == len(dets[i][1]), (conf, i, len(ref[1]), len(dets[i][1]))
            assert np.array_equal(ref[0], dets[i][0]) and np.array_equal(ref[1], dets[i][1]) and np.array_equal(ref[2], dets[i][2]), (conf, i)
    assert max(seen) == 3380 and min(seen) < 400, seen          # the sweep really covers dense and sparse images
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["tail", "lds_overflow", "list_overflow"])
def test_nms_edge_list_limits(case):
    """The suppressing-pair lists past their fast paths: more than 7168 pairs per image (the rounds keep the rest in LDS),
    more than 28 672 (the sorted walk takes over), more than the global list holds (pairs_kernel flags the image).  Crafted
    int8 prediction maps: huge tw / th make every chosen anchor's box the whole image, so all chosen anchors of a class
    suppress each other; expected = the oracle's greedy NMS (one survivor per class, ties to the lower anchor index)."""
    from yolo355.engine import Engine
    C, A, sa_pred = 2, 5, 2
    size = 416 if case == "list_overflow" else 208
    hs = size // 16
    rng = np.random.RandomState(7)
    pq = np.zeros((2, A * (5 + C), hs, hs), np.int8)
    pq[:, :A] = -120                                                  # objectness: sigmoid(-30) -> below any threshold
    n_on = {"tail": 220, "lds_overflow": 845, "list_overflow": 3380}[case]
    for b in range(2):
        flat = rng.permutation(hs * hs * A)[:n_on]                    # anchors switched on
        cell, a = flat // A, flat % A
        y, x = cell // hs, cell % hs
        pq[b, a, y, x] = rng.randint(20, 127, size=n_on)              # objectness in (0.99.., 1): many equal scores
        cls1 = rng.rand(n_on) < (0.0 if case == "tail" else 0.5)      # "tail": one class -> 220^2 / 2 = 24 k pairs
        pq[b, A + a * C + 0, y, x] = np.where(cls1, -100, 100)
        pq[b, A + a * C + 1, y, x] = np.where(cls1, 100, -100)
        pq[b, (1 + C) * A + a * 4 + 2, y, x] = 127                    # tw, th: exp(31.75) x anchor -> clipped to the image
        pq[b, (1 + C) * A + a * 4 + 3, y, x] = 127
    eng = Engine([size, size], C, synth.ANCHOR_SIZE_MASK, conf_thresh=0.5, nms_thresh=0.5, max_batch=2)
    dets = eng.head_nms(pq, sa_pred)
    box, sc = O.head_decode(pq.astype(np.float32) * np.float32(2.0 ** -sa_pred), [size, size], synth.ANCHOR_SIZE_MASK, C)
    for b in range(2):
        ref = O.postprocess(box[b], sc[b], 0.5, 0.5, C)[:3]
        assert 1 <= len(ref[1]) <= C
        assert len(dets[b][1]) == len(ref[1]), (case, b, len(dets[b][1]), len(ref[1]))
        assert np.array_equal(dets[b][2], ref[2])
        assert np.allclose(dets[b][0], ref[0], atol=2e-5, rtol=0) and np.allclose(dets[b][1], ref[1], atol=2e-6, rtol=1e-5)
    eng.close()


@pytest.mark.gpu
def test_kernel_timestamps_of_the_ring_layers():
    """y355_profile(h, 2) / y355_profile_kernels_get: every launch of a forward (the fused front end in slot 0, conv3_1 -> conv3_2 + pool3
    in slot 2, the six other layers, the four head / NMS kernels) reports its own duration, positive and no longer than the interval between the events around
    it; results are unchanged."""
    from yolo355.engine import Engine
    B = 4
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2))
    eng = Engine([416, 416], 2, synth.ANCHOR_SIZE_MASK, max_batch=B)
    eng.load_quantized(ql)
    eng.calibrate(synth.make_images(1, 1, 416, 416), [prep.RangeTracker() for _ in range(11)])
    x = synth.make_images(3, B, 416, 416)
    ref = eng.forward(x)
    eng.profile(2)
    got = eng.forward(x)
    ms, kms, allk = eng.profile_ms(), eng.profile_kernel_ms(), eng.profile_kernels_ms()
    eng.profile(False)
    for a, b in zip(ref, got):
        for u, v in zip(a, b):
            assert np.array_equal(u, v)
    assert len(allk) == 14 and allk[:10] == kms
    # slot 1: conv2 runs inside the fused front end; slot 3: conv3_2 inside the fused pair (slot 2)
    assert kms[1] == 0 and kms[3] == 0 and all(k > 0 for i, k in enumerate(allk) if i not in (1, 3)), allk
    assert all(kms[i] <= ms[i] * 1.25 + 5e-3 for i in range(2, 10)), (kms, ms)        # same forward: the interval contains the kernel
    assert allk[10] + allk[11] <= ms[10] * 1.25 + 5e-3 and allk[12] + allk[13] <= ms[11] * 1.25 + 5e-3, (allk, ms)
    eng.close()


@pytest.mark.gpu
def test_ring_workgroups_option_changes_nothing_but_the_schedule():
    """Y355_OPT_RING_WORKGROUPS: fewer persistent workgroups per launch of the deep convolutions walk more tiles each -- same maps,
    same detections (B = 6 at 416 x 416: 24-48 tiles per layer on 7 workgroups)."""
    from yolo355 import _ffi
    from yolo355.engine import Engine
    B = 6
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2))
    eng = Engine([416, 416], 2, synth.ANCHOR_SIZE_MASK, max_batch=B)
    eng.load_quantized(ql)
    eng.calibrate(synth.make_images(1, 1, 416, 416), [prep.RangeTracker() for _ in range(11)])
    x = synth.make_images(11, B, 416, 416)
    ref = eng.forward(x)
    with pytest.raises(_ffi.Y355Error):                       # the fused front end keeps conv1's map on chip
        eng.get_feature(0, B)
    with pytest.raises(_ffi.Y355Error):                       # ... and the fused conv3_1 -> conv3_2 + pool keeps conv3_1's in LDS (round 5)
        eng.get_feature(2, B)
    taps = [k for k in range(1, 10) if k != 2]
    maps = {k: eng.get_feature(k, B).copy() for k in taps}
    for n in (7, 192):
        eng.set_option(_ffi.OPT_RING_WORKGROUPS, n)
        got = eng.forward(x)
        for k in taps:
            assert np.array_equal(eng.get_feature(k, B), maps[k]), (n, k)
        for a, b in zip(ref, got):
            for u, v in zip(a, b):
                assert np.array_equal(u, v)
    with pytest.raises(_ffi.Y355Error):
        eng.set_option(_ffi.OPT_RING_WORKGROUPS, -1)
    eng.close()


@pytest.mark.gpu
def test_three_handles_throughput_mode_is_exact():
    """bench.py's default configuration -- three engine handles on three HIP streams, deep convolutions on 192 persistent
    workgroups per launch (Y355_OPT_RING_WORKGROUPS), steps alternating without a synchronisation in between: every output
    equals the stand-alone result of a handle that has the GPU to itself (extends test_two_engines_concurrent to the
    round-2 scheduling knob and NMS kernels)."""
    from yolo355 import _ffi
    from yolo355.engine import Engine
    B = 64
    dev = torch.device("cuda", 0)
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2))
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    engs = []
    for st in streams:
        with torch.cuda.stream(st):
            e = Engine([416, 416], 2, synth.ANCHOR_SIZE_MASK, max_batch=B, device=dev)
            e.load_quantized(ql)
        engs.append(e)
    sa = engs[0].calibrate(synth.make_images(1, 1, 416, 416), [prep.RangeTracker() for _ in range(11)])
    for e in engs:
        e.set_act_exponents(sa)
    x = torch.from_numpy(synth.make_images(1000, B, 416, 416)).to(dev)
    bufs = [tuple(torch.zeros_like(t) for t in engs[0]._buffers(B)) for _ in range(7)]
    torch.cuda.synchronize()
    engs[0].forward_device(x, 0, bufs[6])
    torch.cuda.synchronize()
    ref = [t.clone() for t in bufs[6]]
    assert int(ref[3].sum()) > 0
    for e in engs:
        e.set_option(_ffi.OPT_RING_WORKGROUPS, 192)
    for it in range(40):
        for i in range(6):
            with torch.cuda.stream(streams[i % 3]):
                engs[i % 3].forward_device(x, 0, bufs[i])
        torch.cuda.synchronize()
        for i in range(6):
            assert all(torch.equal(u, v) for u, v in zip(ref, bufs[i])), (it, i)
    for e in engs:
        e.close()


@pytest.mark.gpu
def test_capped_pack_kernel_equals_the_torch_packing():
    """y355_pack_dets_capped (one launch, the per-step gather of bench.py) against shard.pack_detections (torch ops, the form the
    gloo tests pin) on the first `cap` detections of every image: byte-identical records, also with padding records."""
    from yolo355 import shard
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    n, md = 5, 300
    boxes = torch.rand((n, md, 4), generator=g).to(dev)
    scores = torch.rand((n, md), generator=g).to(dev)
    cls = torch.randint(0, 20, (n, md), generator=g, dtype=torch.int32).to(dev)
    count = torch.tensor([0, 7, 300, 255, 256], dtype=torch.int32, device=dev)
    for cap, records in ((256, 5), (256, 8), (300, 6), (17, 5)):
        got = torch.full((records, shard.record_bytes(cap)), 0xAB, dtype=torch.uint8, device=dev)
        shard.pack_detections_kernel(boxes, scores, cls, count, records, got, cap)
        want = shard.pack_detections(boxes[:, :cap].contiguous(), scores[:, :cap].contiguous(), cls[:, :cap].contiguous(),
                                     count, records)              # header word 1 keeps the image's own count
        torch.cuda.synchronize()
        assert torch.equal(got, want), (cap, records)
        assert shard.truncated_images(got) == int((count > cap).sum().item())
