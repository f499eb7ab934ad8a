"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle and the golden vectors.

Bit-exact for everything integer (feature maps, pred tensor, exponents, counters);
tolerance for the fp32 head: boxes 2e-5 absolute (normalised coordinates), scores 2e-6
absolute / 1e-5 relative; detection lists are compared tie-tolerantly (helpers.dets_match).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import yolo_oracle as O          # checker only
from yolo355 import synth
from yolo355.engine import Engine, conv3x3_i8_fused
from yolo355.prep import RangeTracker
from helpers import crc, dets_match
from cases import E2E

BOX_TOL, SCORE_TOL = 2e-5, 2e-6


def _same_list(msg):
    """dets_match verdicts that mean "the same detections in the same order": identical lists, or identical boxes / classes
    with every score inside the tolerance (the strict pass also bounds every box coordinate by BOX_TOL; a coordinate a hair
    past it that still rounds to the same grid cell reports "same boxes").  What is NOT accepted here is the tie-group
    fallback: oracle and engine order ties identically by definition."""
    return msg in ("exact", "empty") or msg.startswith("same boxes")
def _rand_i8(seed, shape):
    return (synth.uniform_u8(seed, shape).astype(np.int32) - 128).clip(-127, 127)


def _oracle_layer(q_in, q_w, q_b, sa_in, e_w, e_b, sa_out, leaky, pool):
    t, Fx, _ = O.conv_layer_int(q_in, q_w, q_b, sa_in, e_w, e_b, leaky)
    q = O.rne_shift(t, Fx - sa_out)
    # pooled layers take the 2x2 max FIRST and clamp (and count) on the pooled map -- the max commutes with the monotone
    # requantisation, DESIGN.md section 2 -- so the count is the number of POOLED outputs outside +-127
    if pool:
        q = O.maxpool2x2(q)
    nsat = int((np.abs(q) > 127).sum())
    q = np.clip(q, -127, 127)
    return q.astype(np.int8), int(np.abs(t).max()), Fx, nsat


def test_single_layer_golden(golden):
    """the reference's own Conv2d_fuse + tracker outputs (G1) through y355_conv3x3_i8_fused"""
    n = 0
    while "layer/%d/meta" % n in golden:
        cin, cout, h, w, sa_in, e_w, e_b, sa_out, leaky, s0, s1, s2 = [int(v) for v in golden["layer/%d/meta" % n]]
        q_in, q_w, q_b = _rand_i8(s0, (2, cin, h, w)), _rand_i8(s1, (cout, cin, 3, 3)), _rand_i8(s2, (cout,))
        out, st = conv3x3_i8_fused(q_in, q_w, q_b, sa_in, e_w, e_b, sa_out, leaky=bool(leaky))
        ref = golden["layer/%d/q_out" % n]
        assert np.array_equal(out.astype(np.int32), np.clip(ref, -127, 127)), n
        assert st["saturated"] == int((np.abs(ref) > 127).sum())
        assert np.float32(st["absmax_t"]) * np.float32(2.0 ** -st["frac_bits"]) == golden["layer/%d/ymax" % n][0]
        n += 1
    assert n == 6


@pytest.mark.parametrize("cin,cout,h,w,leaky,pool", [
    (3, 16, 10, 14, True, True), (16, 32, 16, 52, True, True), (24, 40, 9, 7, True, False),
    (32, 64, 13, 26, True, False), (64, 64, 26, 26, True, True), (64, 128, 13, 26, True, False),
    (128, 128, 26, 26, True, True), (128, 256, 13, 13, True, False), (256, 256, 13, 13, True, False),
    (256, 35, 13, 13, False, False), (256, 125, 5, 3, False, False), (200, 70, 6, 10, True, True),
    (1, 1, 1, 1, True, False), (16, 16, 2, 2, False, True),
])
def test_single_layer_shapes(cin, cout, h, w, leaky, pool):
    seed = cin * 1000 + cout
    q_in, q_w, q_b = _rand_i8(seed, (2, cin, h, w)), _rand_i8(seed + 1, (cout, cin, 3, 3)), _rand_i8(seed + 2, (cout,))
    sa_in, e_w, e_b = 5, 8, 6
    amax = 127 * 127 * 9 * cin
    sa_out = int(np.floor(np.log2(127.0 / (amax / 2.0 ** (sa_in + e_w))))) + 2   # some saturation on purpose
    ref, tmax, Fx, nsat = _oracle_layer(q_in, q_w, q_b, sa_in, e_w, e_b, sa_out, leaky, pool)
    out, st = conv3x3_i8_fused(q_in, q_w, q_b, sa_in, e_w, e_b, sa_out, leaky=leaky, pool=pool)
    assert np.array_equal(out, ref)
    assert (st["absmax_t"], st["frac_bits"]) == (tmax, Fx)
    assert st["saturated"] == nsat                       # pooled shapes too: counted on the pooled map


@pytest.mark.parametrize("sa_in,e_w,e_b,sa_out", [(2, 3, 22, 5), (6, 9, 2, 3), (0, 0, 30, 20), (7, 12, 12, 30)])
def test_single_layer_wide_epilogue(sa_in, e_w, e_b, sa_out):
    """exponent gaps that overflow the 32-bit epilogue take the 64-bit kernels (bit-exact too)"""
    q_in, q_w, q_b = _rand_i8(11, (1, 64, 9, 12)), _rand_i8(12, (48, 64, 3, 3)), _rand_i8(13, (48,))
    for leaky, pool in ((True, False), (False, False)):
        ref, tmax, Fx, nsat = _oracle_layer(q_in, q_w, q_b, sa_in, e_w, e_b, sa_out, leaky, pool)
        out, st = conv3x3_i8_fused(q_in, q_w, q_b, sa_in, e_w, e_b, sa_out, leaky=leaky, pool=pool)
        assert np.array_equal(out, ref)
        assert (st["absmax_t"], st["frac_bits"], st["saturated"]) == (tmax, Fx, nsat)


def _build(tag, golden, max_batch=1, max_det=0):
    wkw, anchors, pattern = E2E[tag]
    meta = [int(v) for v in golden[tag + "/meta"]]
    H, W, C, calib_seed = meta[:4]
    confs = [float(v) for v in golden[tag + "/confs"]]
    ql = O.quantize_layers(synth.make_weights(**wkw, num_classes=C))
    eng = Engine([H, W], C, anchors, conf_thresh=confs[0], nms_thresh=0.5, max_batch=max_batch, max_det=max_det)
    eng.load_quantized(ql)
    return eng, ql, (H, W, C, calib_seed, meta[4:], confs, anchors, pattern)


@pytest.mark.parametrize("tag", list(E2E))
def test_end_to_end(golden, tag):
    eng, ql, (H, W, C, calib_seed, img_seeds, confs, anchors, pattern) = _build(tag, golden, max_batch=max(1, len(E2E)))
    xc = synth.make_images(calib_seed, 1, H, W, pattern)
    trackers = [RangeTracker() for _ in range(11)]
    sa = eng.calibrate(xc, trackers, freeze=True)              # first-call semantics (:25-27)
    assert sa == [int(v) for v in golden[tag + "/sa"]]
    otr = [O.RangeTracker() for _ in range(11)]
    r = O.detect(xc, ql, otr, [H, W], anchors, C, confs[0], 0.5, keep=True)
    for li in range(10):
        got = eng.get_feature(li, 1)
        assert np.array_equal(got, r["maps"][li].astype(np.int8)), (tag, li)
        assert crc(got) == int(golden[tag + "/calib/map_crc/%d" % (li + 1)][0]), (tag, li)
    assert np.array_equal(eng.get_feature(9, 1), golden[tag + "/calib/pred_q"])
    modes = []
    for ci, conf in enumerate(confs):
        eng.set_thresholds(conf, 0.5)
        dets = eng.forward(xc, find=(tag == "find"), tap=True)
        assert eng.counters() == (0, 0)
        cb, cs, cc = eng.candidates(1)
        assert np.allclose(cb[0], r["box"][0], atol=BOX_TOL, rtol=0)
        assert np.allclose(cs[0], r["cls_scores"][0].max(1), atol=SCORE_TOL, rtol=1e-5)
        ref = (golden[tag + "/calib/det%d/boxes" % ci], golden[tag + "/calib/det%d/scores" % ci],
               golden[tag + "/calib/det%d/cls" % ci])
        ok, msg = dets_match(ref, dets[0], BOX_TOL, SCORE_TOL, all_scores=r["cls_scores"][0].max(1))
        assert ok and (tag != "diverse" or msg == "exact"), (tag, conf, msg)
        # against the oracle with the same (score desc, index asc) tie order: strict
        ob, os_, oc, _ = O.postprocess(r["box"][0], r["cls_scores"][0], conf, 0.5, C)
        ok, msg = dets_match((ob, os_, oc), dets[0], BOX_TOL, SCORE_TOL, all_scores=r["cls_scores"][0].max(1))
        # the oracle orders ties like the engine (score desc, anchor index asc): identical lists, no tie tolerance (VERDICT r2
        # item 7); tie tolerance is kept for the reference's goldens only (its argsort()[::-1] is unstable)
        assert ok and _same_list(msg), (tag, conf, "vs oracle", msg)
        modes.append(msg.split(":")[0])
    # frozen trackers, whole batch at once == the reference run one image at a time (G7)
    eng.set_thresholds(confs[0], 0.5)
    xs = np.concatenate([synth.make_images(s, 1, H, W, pattern) for s in img_seeds])
    dets = eng.forward(xs)
    rb = O.detect(xs, ql, otr, [H, W], anchors, C, confs[0], 0.5, saturate=True, keep=True)
    assert np.array_equal(eng.get_feature(9, len(img_seeds)), rb["pred_q"].astype(np.int8))
    sat, _ = eng.counters()
    for si in range(len(img_seeds)):
        ref = (golden[tag + "/img%d/boxes" % si], golden[tag + "/img%d/scores" % si], golden[tag + "/img%d/cls" % si])
        if int(golden[tag + "/img%d/nover" % si].sum()) == 0:
            ok, msg = dets_match(ref, dets[si], BOX_TOL, SCORE_TOL, all_scores=rb["cls_scores"][si].max(1))
            assert ok, (tag, si, msg)
        ok, msg = dets_match(rb["dets"][si][:3], dets[si], BOX_TOL, SCORE_TOL, all_scores=rb["cls_scores"][si].max(1))
        assert ok and _same_list(msg), (tag, si, "vs oracle", msg)
        modes.append(msg.split(":")[0])
    print("%s: comparison modes vs the oracle: %s" % (tag, modes))       # pytest -s / -rP shows how many were exact
    eng.close()


def test_guard_trips_like_reference(golden):
    H, W, C, seed, gain = [int(v) for v in golden["guard/meta"]]
    ql = O.quantize_layers(synth.make_weights(seed=2, weight_gain=float(gain), num_classes=C))
    eng = Engine([H, W], C, synth.ANCHOR_SIZE_MASK)
    eng.load_quantized(ql)
    x = synth.make_images(seed, 1, H, W)
    eng.calibrate(x, [RangeTracker() for _ in range(11)])
    eng.forward(x)                       # plain -q path does not look at the guard
    with pytest.raises(AssertionError):
        eng.forward(x, find=True)        # models/slim_yolo_v2.py:222-227
    eng.close()


def test_saturation_is_counted_not_silent():
    """Input 1.6x the calibration range: the reference has no clamp (:35); the engine clamps to
    +-127 like the FPGA and reports how often (SURVEY 7 'hard parts')."""
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2))
    eng = Engine([96, 96], 2, synth.ANCHOR_SIZE_MASK)
    eng.load_quantized(ql)
    x = synth.make_images(5, 1, 96, 96)
    tr = [RangeTracker() for _ in range(11)]
    eng.calibrate(x, tr)
    otr = [O.RangeTracker() for _ in range(11)]
    O.detect(x, ql, otr, [96, 96], synth.ANCHOR_SIZE_MASK, 2)
    x2 = x * np.float32(1.6)
    eng.forward(x2)
    r = O.detect(x2, ql, otr, [96, 96], synth.ANCHOR_SIZE_MASK, 2, saturate=True, keep=True)
    assert np.array_equal(eng.get_feature(9, 1), r["pred_q"].astype(np.int8))
    sat, _ = eng.counters()
    assert sat > 0
    # per layer the engine's count is the oracle's, taken on the layer's output map (pooled layers pool before they clamp);
    # this walks the cold passes of the fused front end and of the convpx kernels as well as the integer epilogues
    got = [eng.layer_stats(k)["saturated"] for k in range(10)]
    print("saturated per layer", got, "oracle", r["sat_out"], "input", r["sat"][0])
    assert got[2:] == r["sat_out"][2:], (got, r["sat_out"])
    assert got[1] == r["sat_out"][1] and got[0] in (r["sat_out"][0], r["sat_out"][0] + r["sat"][0]), (got, r["sat_out"], r["sat"][0])
    assert sat == sum(got) or sat == sum(got) + r["sat"][0], (sat, got)
    assert any(got[k] > 0 for k in (2, 3, 4, 5)), got
    eng.close()


def test_head_only_matches_oracle(golden):
    for tag in ("diverse", "sparse", "gap"):
        wkw, anchors, pattern = E2E[tag]
        H, W, C = [int(v) for v in golden[tag + "/meta"][:3]]
        conf = float(golden[tag + "/confs"][0])
        sa_pred = int(golden[tag + "/sa"][10])
        pq = golden[tag + "/calib/pred_q"]
        eng = Engine([H, W], C, anchors, conf_thresh=conf, max_batch=2)
        dets = eng.head_nms(np.concatenate([pq, pq[:, :, ::-1, :]]), sa_pred)
        cb, cs, cc = eng.candidates(2)
        for i, p in enumerate((pq, pq[:, :, ::-1, :])):
            box, sc = O.head_decode(p.astype(np.float32) * np.float32(2.0 ** -sa_pred), [H, W], anchors, C)
            assert np.allclose(cb[i], box[0], atol=BOX_TOL, rtol=0)
            assert np.allclose(cs[i], sc[0].max(1), atol=SCORE_TOL, rtol=1e-5)
            ref = O.postprocess(box[0], sc[0], conf, 0.5, C)[:3]
            ok, msg = dets_match(ref, dets[i], BOX_TOL, SCORE_TOL, all_scores=sc[0].max(1))
            assert ok and _same_list(msg), (tag, i, msg)     # engine vs oracle: the same tie order by definition
        eng.close()


def test_full_batch_properties():
    """BASELINE size (B=64, 416x416): size-independent properties.  Images repeat with period 4,
    so detections must repeat; element 0..3 are checked against the oracle; the 64-image batch
    equals 4 runs of 16 (batch independence)."""
    B = 64
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2))
    eng = Engine([416, 416], 2, synth.ANCHOR_SIZE_MASK, max_batch=B)
    eng.load_quantized(ql)
    base = synth.make_images(0, 4, 416, 416)
    xc = synth.make_images(1, 1, 416, 416)
    eng.calibrate(xc, [RangeTracker() for _ in range(11)])
    otr = [O.RangeTracker() for _ in range(11)]
    O.detect(xc, ql, otr, [416, 416], synth.ANCHOR_SIZE_MASK, 2)
    x = np.concatenate([base] * (B // 4))
    dets = eng.forward(x)
    pred = eng.get_feature(9, B)
    for i in range(4, B):
        assert np.array_equal(pred[i], pred[i % 4])
        for a, b in zip(dets[i], dets[i % 4]):
            assert np.array_equal(a, b)
    r = O.detect(base, ql, otr, [416, 416], synth.ANCHOR_SIZE_MASK, 2, saturate=True)
    assert np.array_equal(pred[:4], r["pred_q"].astype(np.int8))
    for i in range(4):
        ok, msg = dets_match(r["dets"][i][:3], dets[i], BOX_TOL, SCORE_TOL, all_scores=r["cls_scores"][i].max(1))
        assert ok and _same_list(msg), (i, msg)      # oracle vs engine: same tie order by definition
    d16 = eng.forward(x[:16])
    for i in range(16):
        for a, b in zip(d16[i], dets[i]):
            assert np.array_equal(a, b)
    eng.close()


def test_two_engines_concurrent():
    """Two engine handles on two HIP streams, forwards interleaved (bench.py's configuration): every
    result must equal the handle's stand-alone result.  Regression test for the LDS-DMA restage hazard
    of the production conv kernel, which only showed under memory contention from the other stream
    (rare wrong tiles of the prediction layer)."""
    import torch
    B = 64
    dev = torch.device("cuda", 0)
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2))
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    engs = []
    for st in streams:
        with torch.cuda.stream(st):
            e = Engine([416, 416], 2, synth.ANCHOR_SIZE_MASK, max_batch=B, device=dev)
            e.load_quantized(ql)
        engs.append(e)
    sa = engs[0].calibrate(synth.make_images(1, 1, 416, 416), [RangeTracker() for _ in range(11)])
    for e in engs:
        e.set_act_exponents(sa)
    x = torch.from_numpy(synth.make_images(1000, B, 416, 416)).to(dev)
    # zero-filled outputs: entries past count[b] are never written, so whole tensors compare equal
    bufs = [tuple(torch.zeros_like(t) for t in engs[0]._buffers(B)) for _ in range(3)]
    torch.cuda.synchronize()

    def same(a, b):
        return all(torch.equal(u, v) for u, v in zip(a, b))
    engs[0].forward_device(x, 0, bufs[0])
    torch.cuda.synchronize()
    ref = [t.clone() for t in bufs[0]]
    pred_ref = engs[0].get_feature(9, B).copy()
    assert int(ref[3].sum()) > 0
    engs[1].forward_device(x, 0, bufs[1])
    torch.cuda.synchronize()
    assert same(ref, bufs[1])
    for it in range(300):
        engs[0].forward_device(x, 0, bufs[0])
        engs[1].forward_device(x, 0, bufs[1])
        engs[0].forward_device(x, 0, bufs[2])
        torch.cuda.synchronize()
        if not (same(ref, bufs[1]) and same(ref, bufs[2])):
            diff = int((engs[1].get_feature(9, B) != pred_ref).sum())
            raise AssertionError("iteration %d: concurrent result differs (%d prediction values)" % (it, diff))
    for e in engs:
        e.close()


def test_forward_frames_equals_normalised_tensor():
    """y355_forward_u8 (uint8 HWC BGR frames, BaseTransform fused into the first layer) == y355_forward on
    the tensor the reference's BaseTransform + channel swap + permute produce (data/__init__.py:30-56,
    test.py:79): identical int8 feature maps and detections, fused path and fp32 staging path (find)."""
    B, H, W = 3, 224, 320
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2, pred_gain=400.0, obj_bias=-4.0))
    eng = Engine([H, W], 2, synth.ANCHOR_SIZE_MASK, max_batch=B)
    eng.load_quantized(ql)
    frames = synth.make_frames_u8(5, B, H, W, "blocks")
    x = synth.normalize_frames(frames)
    eng.calibrate(x[:1], [RangeTracker() for _ in range(11)])
    ref = eng.forward(x)
    ctr_ref = eng.counters()
    feat_ref = [eng.get_feature(k, B).copy() for k in (1, 9)]
    got = eng.forward_frames(frames)
    assert eng.counters() == ctr_ref                       # clamped inputs / outputs counted identically
    for k, f in zip((1, 9), feat_ref):
        assert np.array_equal(eng.get_feature(k, B), f)
    for a, b in zip(ref, got):
        for u, v in zip(a, b):
            assert np.array_equal(u, v)
    got2 = eng.forward_frames(frames, find=True)          # guard flag: normalise kernel + fp32 path
    for a, b in zip(ref, got2):
        for u, v in zip(a, b):
            assert np.array_equal(u, v)
    eng.close()


@pytest.mark.parametrize("H,W,B,gain", [(112, 64, 1, 2.5), (320, 416, 2, 2.5), (240, 320, 2, 3.0), (96, 96, 2, 4.0)])
def test_small_map_saturation_counts(H, W, B, gain):
    """Per-layer saturation counts on maps that are NOT multiples of the kernels' tiles (7 x 4 ... 20 x 26 behind conv4_2): the rows and
    columns of an edge tile that lie beyond the map are computed from the halo and from whatever is behind it, they are not
    outputs and must not be counted when they clamp (round 5: the ring kernels counted them -- conv5 22 against the oracle's 20 at
    112 x 64)."""
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2, pred_gain=400.0, obj_bias=-4.0))
    eng = Engine([H, W], 2, synth.ANCHOR_SIZE_MASK, max_batch=B)
    eng.load_quantized(ql)
    frames = synth.make_frames_u8(11, B, H, W, "blocks")
    xc = synth.normalize_frames(frames)[:1]
    eng.calibrate(xc, [RangeTracker() for _ in range(11)])
    otr = [O.RangeTracker() for _ in range(11)]
    O.detect(xc, ql, otr, [H, W], synth.ANCHOR_SIZE_MASK, 2)
    x = synth.normalize_frames(frames) * np.float32(gain)
    r = O.detect(x, ql, otr, [H, W], synth.ANCHOR_SIZE_MASK, 2, saturate=True, keep=True)
    eng.forward(x)
    got = [eng.layer_stats(k)["saturated"] for k in range(10)]
    print("saturated per layer", got, "oracle", list(r["sat_out"]), "input", r["sat"][0])
    assert got[1:] == list(r["sat_out"])[1:], (got, r["sat_out"])
    assert got[0] in (r["sat_out"][0], r["sat_out"][0] + r["sat"][0]), (got, r["sat_out"], r["sat"][0])
    assert np.array_equal(eng.get_feature(9, B), r["pred_q"].astype(np.int8))
    assert sum(got[4:]) > 0, "the fixture must clamp in the deep layers"
    eng.close()


def test_deep_layers_clamp_on_the_ring_kernels():
    """The deep layers' fp32 epilogue (csrc/conv3x3_ring.hip) stages unclamped bytes in its hot pass and re-stages a tile
    clamped when a value left [-127, 127]: with the output exponents of conv5 .. pred raised by two (every activation four
    times larger) the maps, the prediction and the per-layer saturation counts must still equal the oracle's."""
    H = W = 416
    B = 2
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2, pred_gain=400.0, obj_bias=-4.0))
    eng = Engine([H, W], 2, synth.ANCHOR_SIZE_MASK, max_batch=B)
    eng.load_quantized(ql)
    frames = synth.make_frames_u8(11, B, H, W, "blocks")
    xc = synth.normalize_frames(frames)[:1]
    eng.calibrate(xc, [RangeTracker() for _ in range(11)])
    otr = [O.RangeTracker() for _ in range(11)]
    O.detect(xc, ql, otr, [H, W], synth.ANCHOR_SIZE_MASK, 2)
    sa = list(eng.get_act_exponents())
    for k in (7, 8, 9, 10):                      # trackers behind conv5, conv6, conv7, pred
        sa[k] += 2
        otr[k].scale = otr[k].scale * 4
    eng.set_act_exponents(sa)
    x = synth.normalize_frames(frames)
    r = O.detect(x, ql, otr, [H, W], synth.ANCHOR_SIZE_MASK, 2, saturate=True, keep=True)
    assert list(r["sa"]) == sa
    eng.forward(x)
    sat = [eng.layer_stats(k)["saturated"] for k in range(10)]
    assert sat == list(r["sat_out"]), (sat, r["sat_out"])
    assert min(sat[6:10]) > 0, "the fixture must clamp in every ring-kernel layer"
    for k in (6, 7, 8):
        assert np.array_equal(eng.get_feature(k, B), r["maps"][k].astype(np.int8)), "layer %d" % k
    assert np.array_equal(eng.get_feature(9, B), r["pred_q"].astype(np.int8))
    eng.close()


@pytest.mark.parametrize("H,W,B,gain,bump", [(416, 416, 3, 1.0, 0), (96, 96, 2, 1.0, 0), (320, 416, 2, 1.0, 0), (240, 320, 2, 1.6, 0),
                                              (112, 64, 1, 2.5, 0), (320, 608, 1, 1.0, 0), (416, 416, 2, 3.0, 0), (416, 416, 2, 1.0, 2),
                                              (64, 416, 1, 1.0, 1), (416, 128, 5, 1.0, 0)])
def test_fused_pairs_equal_layer_launches(H, W, B, gain, bump):
    """conv3_1 -> conv3_2 + pool3 as ONE launch (csrc/pxpair.hip, the default; conv3_1's map never leaves LDS) against the
    oracle's conv3_2 map and against the one-launch-per-layer route: identical int8 maps behind the pair, detections, and
    PER-LAYER saturation counts (conv3_1's counted once although a band recomputes its neighbours' edge rows).  Sizes: bands
    that start and end inside an image (B x H/8 pooled rows over 256 workgroups), maps narrower than a 16-pixel group row
    allows (W / 4 = 16), a map too wide for the rings (W = 608: the launcher declines and the two layers run as before),
    inputs beyond the calibration range, and `bump`: the exponents behind conv3_1 and conv3_2 raised so that BOTH phases take
    their cold (clamping, counting) passes."""
    from yolo355 import _ffi
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2, pred_gain=400.0, obj_bias=-4.0))
    eng = Engine([H, W], 2, synth.ANCHOR_SIZE_MASK, max_batch=B)
    eng.load_quantized(ql)
    frames = synth.make_frames_u8(11, B, H, W, "blocks")
    xc = synth.normalize_frames(frames)[:1]
    eng.calibrate(xc, [RangeTracker() for _ in range(11)])
    otr = [O.RangeTracker() for _ in range(11)]
    O.detect(xc, ql, otr, [H, W], synth.ANCHOR_SIZE_MASK, 2)
    if bump:
        sa = list(eng.get_act_exponents())
        for k in (3, 4, 5, 6):                   # trackers behind conv3_1, conv3_2, conv4_1, conv4_2
            sa[k] += bump
            otr[k].scale = otr[k].scale * 2 ** bump
        eng.set_act_exponents(sa)
    x = synth.normalize_frames(frames) * np.float32(gain)
    r = O.detect(x, ql, otr, [H, W], synth.ANCHOR_SIZE_MASK, 2, saturate=True, keep=True)
    res = {}
    for fuse in (1, 0):                          # 1 (the default): the pair fused, the layers on different waves of every SIMD; 0: two launches
        eng.set_option(_ffi.OPT_FUSE_PAIRS, fuse)
        dets = eng.forward(x)
        res[fuse] = (dets, [eng.layer_stats(k)["saturated"] for k in range(10)], eng.get_feature(3, B).copy(), eng.get_feature(9, B).copy(),
                     eng.get_feature(5, B).copy())
        if fuse == 0:
            assert np.array_equal(eng.get_feature(2, B), r["maps"][2].astype(np.int8))
            assert np.array_equal(eng.get_feature(4, B), r["maps"][4].astype(np.int8))
    for v in (2, 4, 3, -1):                      # round 5's other schedules left the library: the option is 0 / 1
        with pytest.raises(_ffi.Y355Error) as e:
            eng.set_option(_ffi.OPT_FUSE_PAIRS, v)
        assert e.value.code == _ffi.EINVAL
    eng.set_option(_ffi.OPT_FUSE_PAIRS, 1)
    eng.forward(x)
    if W // 4 <= 104:                            # the fused launch ran: conv3_1's map was not written
        with pytest.raises(_ffi.Y355Error) as e:
            eng.get_feature(2, B)
        assert e.value.code == _ffi.ENOTREADY
    else:
        assert np.array_equal(eng.get_feature(2, B), r["maps"][2].astype(np.int8))
    assert np.array_equal(eng.get_feature(4, B), r["maps"][4].astype(np.int8))      # conv4_1 always has its own launch
    assert np.array_equal(res[1][4], r["maps"][5].astype(np.int8)), "conv4_2's pooled map differs from the oracle"
    assert np.array_equal(res[1][2], r["maps"][3].astype(np.int8)), "conv3_2's pooled map of the fused launch differs from the oracle"
    assert np.array_equal(res[1][3], r["pred_q"].astype(np.int8))
    assert np.array_equal(res[1][2], res[0][2]) and np.array_equal(res[1][3], res[0][3]) and np.array_equal(res[1][4], res[0][4])
    assert res[1][1] == res[0][1], "per-layer saturation counts differ between the fused and the per-layer route: %s / %s" % (res[1][1], res[0][1])
    assert res[1][1][2:] == list(r["sat_out"])[2:], (res[1][1], r["sat_out"])
    if bump:
        assert min(res[1][1][2:6]) > 0, "the fixture must clamp in both layers of the pair and in conv4_1 / conv4_2"
    for a, b in zip(res[1][0], res[0][0]):
        for u, v in zip(a, b):
            assert np.array_equal(u, v)
    eng.close()


@pytest.mark.parametrize("H,W,B,gain", [(416, 416, 3, 1.0), (96, 96, 2, 1.0), (320, 416, 2, 1.0), (240, 320, 2, 1.6),
                                         (112, 64, 1, 2.5), (320, 608, 1, 1.0), (416, 416, 2, 3.0)])   # W = 608: input rows wider than 8 DMA pieces (convpx)
def test_fused_front_end_equals_layer_launches(H, W, B, gain):
    """conv1 + pool1 + conv2 + pool2 as ONE launch (csrc/front.hip, the default) against the oracle's conv2 map and against
    the one-launch-per-layer route: identical int8 maps, detections and saturation counters; fp32 tensors and uint8 frames;
    edge tiles (sizes that are not multiples of the 52-pixel tile), inputs beyond the calibration range (clamped pixels and
    outputs must be counted once although halo pixels are computed by two tiles)."""
    from yolo355 import _ffi
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2, pred_gain=400.0, obj_bias=-4.0))
    eng = Engine([H, W], 2, synth.ANCHOR_SIZE_MASK, max_batch=B)
    eng.load_quantized(ql)
    frames = synth.make_frames_u8(11, B, H, W, "blocks")
    xc = synth.normalize_frames(frames)[:1]
    eng.calibrate(xc, [RangeTracker() for _ in range(11)])
    otr = [O.RangeTracker() for _ in range(11)]
    O.detect(xc, ql, otr, [H, W], synth.ANCHOR_SIZE_MASK, 2)
    x = synth.normalize_frames(frames) * np.float32(gain)
    r = O.detect(x, ql, otr, [H, W], synth.ANCHOR_SIZE_MASK, 2, saturate=True, keep=True)
    res = {}
    for fuse in (1, 0):
        eng.set_option(_ffi.OPT_FUSE_FRONT, fuse)
        dets = eng.forward(x)
        res[fuse] = (dets, eng.counters(), eng.get_feature(1, B).copy(), eng.get_feature(9, B).copy())
    assert np.array_equal(res[1][2], r["maps"][1].astype(np.int8)), "conv2 map of the fused launch differs from the oracle"
    assert np.array_equal(res[1][3], r["pred_q"].astype(np.int8))
    assert np.array_equal(res[1][2], res[0][2]) and np.array_equal(res[1][3], res[0][3])
    assert res[1][1] == res[0][1], "saturation counters differ between the fused and the per-layer route"
    if gain > 1.0:
        assert res[1][1][0] > 0
    for a, b in zip(res[1][0], res[0][0]):
        for u, v in zip(a, b):
            assert np.array_equal(u, v)
    if gain == 1.0:
        for fuse in (1, 0):
            eng.set_option(_ffi.OPT_FUSE_FRONT, fuse)
            got = eng.forward_frames(frames)
            assert eng.counters() == res[1][1]
            assert np.array_equal(eng.get_feature(1, B), res[1][2]) and np.array_equal(eng.get_feature(9, B), res[1][3])
            for a, b in zip(res[1][0], got):
                for u, v in zip(a, b):
                    assert np.array_equal(u, v)
    eng.close()


def test_elementwise_operators():
    """stand-alone input fake-quant and 2x2 max-pool against the oracle's restatement"""
    from yolo355.engine import quantize_input_f32_i8, maxpool2x2_i8
    x = synth.make_images(9, 2, 64, 96) * np.float32(3.0)            # some values beyond +-127 / 2^sa
    for sa in (4, 5, 6):
        q, clamped = quantize_input_f32_i8(x, sa)
        r = np.rint(x * np.float32(2.0 ** sa))
        assert np.array_equal(q, np.clip(r, -127, 127).astype(np.int8))
        assert clamped == int((np.abs(r) > 127).sum())
    q, _ = quantize_input_f32_i8(x, 4)
    p = maxpool2x2_i8(q)
    B, C_, H, W = q.shape
    assert np.array_equal(p, q.reshape(B, C_, H // 2, 2, W // 2, 2).max(axis=(3, 5)))


def test_errors_are_loud():
    """wrong shapes, missing weights / exponents, oversize batches: negative status + message, no
    silent result (errors of the reference are Python exceptions)."""
    from yolo355 import _ffi
    from yolo355.engine import maxpool2x2_i8
    eng = Engine([96, 96], 2, synth.ANCHOR_SIZE_MASK, max_batch=2)
    x = synth.make_images(1, 1, 96, 96)
    with pytest.raises(_ffi.Y355Error) as e:
        eng.forward(x)                                  # nothing loaded
    assert e.value.code == -3                           # Y355_ENOTREADY
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2))
    eng.load_quantized(ql)
    with pytest.raises(_ffi.Y355Error) as e:
        eng.forward(x)                                  # not calibrated
    assert e.value.code == -3
    with pytest.raises(_ffi.Y355Error):
        eng.load_layer(3, ql[2]["q_w"], ql[2]["q_b"], 9, 9)      # wrong layer shape
    with pytest.raises(ValueError):
        eng.forward(synth.make_images(1, 3, 96, 96))    # batch > max_batch
    with pytest.raises(ValueError):
        eng.forward(synth.make_images(1, 1, 64, 96))    # wrong input size
    with pytest.raises(_ffi.Y355Error):
        Engine([100, 96], 2, synth.ANCHOR_SIZE_MASK)    # not a multiple of 16
    with pytest.raises(_ffi.Y355Error):
        maxpool2x2_i8(np.zeros((1, 1, 3, 4), np.int8))  # odd height
    eng.close()
