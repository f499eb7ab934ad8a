"""Pins the CPU oracle (oracle/yolo_oracle.py) to outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/gen_golden.py)."""
import numpy as np
import pytest

from oracle import yolo_oracle as O
from yolo355 import synth
from helpers import crc, dets_match


def test_weight_prep_matches_reference(golden):
    for tag, kw in [("w2", dict(seed=2)), ("w3gap", dict(seed=3, bias_gain=40.0, weight_gain=3.0))]:
        ws = synth.make_weights(**kw, num_classes=2)
        ql = O.quantize_layers(ws)
        for li, L in enumerate(ql):
            g = golden["prep/%s/%d" % (tag, li)]
            assert (L["e_w"], L["e_b"]) == (g[0], g[1])
            assert crc(L["q_w"].astype(np.int8)) == g[2]
            assert crc(L["q_b"].astype(np.int8)) == g[3]
            assert np.abs(L["q_w"]).max() == g[4] <= 127


def test_bn_fold_matches_reference(golden):
    for case in range(3):
        cin, cout, with_bias = golden["fuse/%d/meta" % case]
        w = synth.uniform_pm1(100 + case, (cout, cin, 3, 3)) * np.float32(0.2)
        b = synth.uniform_pm1(200 + case, (cout,)) * np.float32(0.3)
        g, be, mu, var = synth.make_bn(300 + case, cout)
        wf, bf = O.fuse_conv_and_bn(w, b if with_bias else None, g, be, mu, var)
        assert np.array_equal(wf, golden["fuse/%d/w" % case])
        assert np.array_equal(bf, golden["fuse/%d/b" % case])


def test_single_layer_matches_reference(golden):
    n = 0
    while "layer/%d/meta" % n in golden:
        cin, cout, h, w, sa_in, e_w, e_b, sa_out, leaky, s0, s1, s2 = [int(v) for v in golden["layer/%d/meta" % n]]
        q_in = (synth.uniform_u8(s0, (2, cin, h, w)).astype(np.int32) - 128).clip(-127, 127)
        q_w = (synth.uniform_u8(s1, (cout, cin, 3, 3)).astype(np.int32) - 128).clip(-127, 127)
        q_b = (synth.uniform_u8(s2, (cout,)).astype(np.int32) - 128).clip(-127, 127)
        t, Fx, _ = O.conv_layer_int(q_in, q_w, q_b, sa_in, e_w, e_b, bool(leaky))
        q = O.rne_shift(t, Fx - sa_out)
        assert np.array_equal(q, golden["layer/%d/q_out" % n].astype(np.int64)), n
        ymax = np.float32(np.abs(t).max()) * np.float32(2.0 ** -Fx)
        assert ymax == golden["layer/%d/ymax" % n][0]
        n += 1
    assert n == 6


from cases import E2E


@pytest.mark.parametrize("tag", list(E2E))
def test_end_to_end_matches_reference(golden, tag):
    wkw, anchors, pattern = E2E[tag]
    meta = [int(v) for v in golden[tag + "/meta"]]
    H, W, C, calib_seed = meta[:4]
    img_seeds = meta[4:]
    confs = [float(v) for v in golden[tag + "/confs"]]
    ql = O.quantize_layers(synth.make_weights(**wkw, num_classes=C))
    trackers = [O.RangeTracker() for _ in range(11)]
    xc = synth.make_images(calib_seed, 1, H, W, pattern)
    r = O.detect(xc, ql, trackers, [H, W], anchors, C, confs[0], 0.5, find=(tag == "find"), keep=True)
    assert r["sa"] == list(golden[tag + "/sa"])
    assert np.allclose(np.float32(r["guard"]) * 0 + 1, 1)
    # per-layer int8 feature maps (tap 0 = quantized input, taps 1..10 = layers)
    for li in range(1, 11):
        g = golden[tag + "/calib/map_crc/%d" % li]
        m = r["maps"][li - 1]
        assert crc(m.astype(np.int8)) == g[0], (tag, li)
    assert np.array_equal(r["pred_q"].astype(np.int8), golden[tag + "/calib/pred_q"])
    assert max(r["sat"]) == 0                      # first call: scale covers the input
    assert max(r["acc_max"]) < 2 ** 24             # fp32-exact regime of the reference
    for ci, conf in enumerate(confs):
        b, s, c, _ = O.postprocess(r["box"][0], r["cls_scores"][0], conf, 0.5, C)
        ref = (golden[tag + "/calib/det%d/boxes" % ci], golden[tag + "/calib/det%d/scores" % ci],
               golden[tag + "/calib/det%d/cls" % ci])
        ok, msg = dets_match(ref, (b, s, c), box_tol=0, score_tol=0, all_scores=r["cls_scores"][0].max(1))
        print(tag, conf, msg)
        assert ok and (tag != "diverse" or msg == "exact"), (tag, conf, msg)
    # frozen-tracker images (batch semantics: element i alone == element i of the batch)
    xs = np.concatenate([synth.make_images(s, 1, H, W, pattern) for s in img_seeds])
    rb = O.detect(xs, ql, trackers, [H, W], anchors, C, confs[0], 0.5, find=(tag == "find"))
    assert rb["sa"] == r["sa"]
    for si in range(len(img_seeds)):
        nover = golden[tag + "/img%d/nover" % si]
        ref = (golden[tag + "/img%d/boxes" % si], golden[tag + "/img%d/scores" % si],
               golden[tag + "/img%d/cls" % si])
        if nover.sum() == 0:
            assert np.array_equal(rb["pred_q"][si].astype(np.int8), golden[tag + "/img%d/pred_q" % si][0])
        ok, msg = dets_match(ref, rb["dets"][si][:3], box_tol=0, score_tol=0, all_scores=rb["cls_scores"][si].max(1))
        print(tag, si, msg)
        assert ok, (tag, si, msg)


def test_guard_trips_like_reference(golden):
    H, W, C, seed, gain = [int(v) for v in golden["guard/meta"]]
    assert golden["guard/tripped"][0] == 1
    ql = O.quantize_layers(synth.make_weights(seed=2, weight_gain=float(gain), num_classes=C))
    trackers = [O.RangeTracker() for _ in range(11)]
    with pytest.raises(AssertionError):
        O.detect(synth.make_images(seed, 1, H, W), ql, trackers, [H, W], synth.ANCHOR_SIZE_MASK, C, find=True)
