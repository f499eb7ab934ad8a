"""CPU tests of the host logic and of the C-ABI surface (no compute calls)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from yolo355 import _ffi, prep, synth
from helpers import crc


def test_library_exports_every_declared_symbol():
    names = _ffi.declared_symbols()
    assert len(names) >= 25 and "y355_forward" in names
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libyolo355.so does not export %s" % n
    assert set(_ffi._SIGS) <= set(names)
    assert lib.y355_version() == 2


def test_create_without_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from yolo355.engine import Engine
    with pytest.raises(RuntimeError):
        Engine([416, 416], 2, synth.ANCHOR_SIZE_MASK)
    lib = _ffi.lib()
    cfg = _ffi.Config()
    cfg.height, cfg.width, cfg.num_classes, cfg.num_anchors, cfg.max_batch = 100, 416, 2, 5, 1
    h = ctypes.c_void_p()
    assert lib.y355_create(ctypes.byref(cfg), ctypes.byref(h)) == -1      # Y355_EINVAL: not a multiple of 16
    assert b"multiple of 16" in lib.y355_last_error()


def test_prep_quantizers_match_golden(golden):
    for tag, kw in [("w2", dict(seed=2)), ("w3gap", dict(seed=3, bias_gain=40.0, weight_gain=3.0))]:
        for li, (name, w, b) in enumerate(synth.make_weights(**kw, num_classes=2)):
            g = golden["prep/%s/%d" % (tag, li)]
            qw, ew = prep.to_int8_pow2(torch.from_numpy(w))
            qb, eb = prep.to_int8_pow2(torch.from_numpy(b))
            assert (ew, eb) == (g[0], g[1])
            assert crc(qw.astype(np.int8)) == g[2] and crc(qb.astype(np.int8)) == g[3]
            # a tensor stored as q/2^e is recognised again, raw fp32 weights are refused
            q2, e2 = prep.as_dyadic_int8(torch.from_numpy(qw.astype(np.float32) / np.float32(2.0 ** ew)))
            assert np.array_equal(q2.astype(np.float64) * 2.0 ** -e2, qw.astype(np.float64) * 2.0 ** -ew)
            with pytest.raises(ValueError):
                prep.as_dyadic_int8(torch.from_numpy(w))


def test_fuse_conv_and_bn_matches_golden(golden):
    for case in range(3):
        cin, cout, with_bias = [int(v) for v in golden["fuse/%d/meta" % case]]
        w = synth.uniform_pm1(100 + case, (cout, cin, 3, 3)) * np.float32(0.2)
        b = synth.uniform_pm1(200 + case, (cout,)) * np.float32(0.3)
        g, be, mu, var = synth.make_bn(300 + case, cout)
        conv = torch.nn.Conv2d(cin, cout, 3, 1, 1, bias=bool(with_bias))
        bn = torch.nn.BatchNorm2d(cout)
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(w))
            if with_bias:
                conv.bias.copy_(torch.from_numpy(b))
            bn.weight.copy_(torch.from_numpy(g)); bn.bias.copy_(torch.from_numpy(be))
            bn.running_mean.copy_(torch.from_numpy(mu)); bn.running_var.copy_(torch.from_numpy(var))
        f = prep.fuse_conv_and_bn(conv, bn)
        assert np.array_equal(f.weight.detach().numpy(), golden["fuse/%d/w" % case])
        assert np.array_equal(f.bias.detach().numpy(), golden["fuse/%d/b" % case])
        # corrected fold == eval-mode conv+bn
        fc = prep.fuse_conv_and_bn(conv, bn, corrected=True)
        x = torch.from_numpy(synth.uniform_pm1(7, (1, cin, 6, 6)))
        bn.eval()
        with torch.no_grad():
            assert torch.allclose(fc(x), bn(conv(x)), atol=1e-5)


def test_range_tracker_state_machine():
    t = prep.RangeTracker()
    assert t.update(np.float32(2.64), True) == 5          # first call calibrates even when frozen (:25-27)
    assert t.update(np.float32(100.0), True) == 5         # frozen afterwards
    e = t.update(np.float32(0.5), False)                  # EMA (:31)
    s = torch.tensor([127 / np.float32(2.64)], dtype=torch.float32) * (1 - 0.1) + (127 / torch.tensor(0.5)) * 0.1
    assert e == int(torch.floor(torch.log2(s)).item())


def test_split_product_requant_is_exact():
    """y355_requant_gen32 with Requant::split (csrc/y355_common.h): the negative branch's t * neg_mul taken as 256 A + rem with
    A = (t >> 8) * neg_mul + ((t & 255) * neg_mul >> 8) and the remainder as a sticky bit equals RNE(t * neg_mul / 2^sh) computed
    in 64 bits -- for the slope 205 / 2048 of the int8 YOLOv3tiny and every shift the host admits (9 .. 31), around the ties."""
    rng = np.random.default_rng(5)
    for nm in (205, 1, 3, 2047):
        for sh in range(9, 32):
            base = rng.integers(-2 ** 30 + 1, 1, 200000)
            k = rng.integers(-2 ** 30 // (1 << sh) * 0, 2 ** 12, 4000)
            ties = -(((2 * k + 1) << (sh - 1)) // nm)[:, None] + np.arange(-3, 4)[None, :]      # t * nm near odd multiples of 2^(sh-1)
            t = np.concatenate([base, ties.ravel(), -np.arange(0, 70000)]).astype(np.int64)
            t = t[(t <= 0) & (t > -2 ** 30)]
            P = t * nm
            ref = (P + (1 << (sh - 1)) - 1 + ((P >> sh) & 1)) >> sh                                # 64-bit RNE shift
            hi, m2 = t >> 8, (t & 255) * nm
            A = hi * nm + (m2 >> 8)
            s = sh - 8
            got = (A + (1 << (s - 1)) - 1 + ((((m2 & 255) + 255) >> 8) | ((A >> s) & 1))) >> s
            assert np.abs(A).max() < 2 ** 31 or nm == 2047                                         # the host's bound for 205 / 2048
            assert np.array_equal(got, ref), (nm, sh)


def test_detection_difference_attribution_separates_noise_from_a_wrong_nms():
    """helpers.explain_detection_differences (the floating-point configs' list-level evidence): per-anchor noise inside a
    tolerance leaves no unexplained difference; an NMS with another threshold on one side does."""
    import helpers
    rng = np.random.default_rng(0)
    N = 1500
    c, wh = rng.uniform(0.1, 0.9, (N, 2)), rng.uniform(0.02, 0.3, (N, 2))
    box = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    score, cls = rng.uniform(0, 0.6, N).astype(np.float32), rng.integers(0, 2, N)
    gb = (box + rng.normal(0, 0.004, box.shape)).astype(np.float32)
    gs = (score + rng.normal(0, 0.005, N)).astype(np.float32)
    gc = cls.copy()
    gc[::97] ^= 1                                          # a few best-class flips
    r = helpers.explain_detection_differences(box, score, cls, gb, gs, gc, 0.01, 0.5)
    assert r["differ"] > 50 and r["unexplained"] == [] and min(r["causes"].values()) > 0, r
    assert r["root_score_dev"] <= 0.03 and r["root_box_dev"] <= 0.03
    same = helpers.explain_detection_differences(box, score, cls, box, score, cls, 0.01, 0.5)
    assert same["differ"] == 0 and same["n_ref"] == same["n_got"]
    orig, calls = helpers._nms_replay, [0]

    def wrong(b, s, k, ct, nt):                            # the second side suppresses at IoU 0.6
        calls[0] += 1
        return orig(b, s, k, ct, 0.6 if calls[0] == 2 else nt)
    helpers._nms_replay = wrong
    try:
        bad = helpers.explain_detection_differences(box, score, cls, gb, gs, cls, 0.01, 0.5)
    finally:
        helpers._nms_replay = orig
    assert len(bad["unexplained"]) > 50


def test_round6_entry_points_reject_bad_arguments_without_a_gpu():
    """argument checks of the round-6 C ABI (pipeline, calibration, device-resident operators) come before any HIP call"""
    import ctypes as C
    from yolo355 import _ffi
    lib = _ffi.lib()
    h = C.c_void_p()
    assert lib.y355_pipeline_create(None, 0, -1, C.byref(h)) == _ffi.EINVAL
    cfg = _ffi.Config()
    cfg.height = cfg.width = 416
    cfg.num_classes, cfg.num_anchors, cfg.max_batch = 2, 5, 1
    assert lib.y355_pipeline_create(C.byref(cfg), 9, -1, C.byref(h)) == _ffi.EINVAL and b"handles" in lib.y355_last_error()
    assert lib.y355_pipeline_create(C.byref(cfg), 2, 5000, C.byref(h)) == _ffi.EINVAL
    streams = (C.c_void_p * 2)()
    assert lib.y355_pipeline_create_on(C.byref(cfg), 0, -1, streams, C.byref(h)) == _ffi.EINVAL      # caller streams need a count
    t = C.c_longlong()
    assert lib.y355_pipeline_submit(None, None, 1, 0, None, None, None, None, None, C.byref(t)) == _ffi.EINVAL
    assert lib.y355_pipeline_wait(None, 0, 0, None) == _ffi.EINVAL and lib.y355_pipeline_handles(None) == _ffi.EINVAL
    assert lib.y355_pipeline_engine(None, 0) is None and lib.y355_pipeline_stream(None, 0) is None
    assert lib.y355_calibrate(None, None, 1, 1, 0.1, None, None) == _ffi.EINVAL
    assert lib.y355_set_trackers(None, None, None) == _ffi.EINVAL and lib.y355_get_trackers(None, None, None) == _ffi.EINVAL
    assert lib.y355_conv_op_create_bf16(0, None, None, 3, 8, 3, 1, 0.1, C.byref(h)) == _ffi.EINVAL
    assert lib.y355_conv_op_create_i8(0, None, None, 3, 8, 0, 0, 0, C.byref(h)) == _ffi.EINVAL
    assert lib.y355_conv_op_forward(None, None, None, 1, 8, 8, 0, None, None) == _ffi.EINVAL
    e = C.c_int32()
    assert lib.y355_conv_op_forward_i8(None, None, 1, 8, 8, None, None, None, C.byref(e)) == _ffi.EINVAL
    assert lib.y355_reorg_f32_dev(None, 1, 1, 4, 4, 2, None, None) == _ffi.EINVAL
    assert lib.y355_spp_f32_dev(None, 1, 1, 4, 4, None, None) == _ffi.EINVAL
    lib.y355_pipeline_destroy(None)                       # no-ops on null
    lib.y355_conv_op_destroy(None)


def test_missing_extension_fails_loudly_everywhere(monkeypatch, tmp_path):
    """no libyolo355.so -> ImportError from every product entry point; nothing computes on the CPU instead"""
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "libyolo355.so"))
    from yolo355.engine import Engine, Pipeline, conv3x3_i8_fused
    from yolo355.utils.modules import Conv2d_fuse
    with pytest.raises(ImportError, match="no CPU fallback"):
        Engine([416, 416], 2, synth.ANCHOR_SIZE_MASK)
    with pytest.raises(ImportError, match="no CPU fallback"):
        Pipeline([416, 416], 2, synth.ANCHOR_SIZE_MASK)
    q = np.zeros((1, 16, 4, 4), np.int32)
    with pytest.raises(ImportError, match="no CPU fallback"):
        conv3x3_i8_fused(q, np.zeros((16, 16, 3, 3), np.int32), np.zeros(16, np.int32), 5, 8, 6, 3)
    m = Conv2d_fuse(16, 16, 3, padding=1, leakyReLU=True).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.round(p * 64) / 64)
        with pytest.raises(ImportError, match="no CPU fallback"):
            m(torch.round(torch.randn(1, 16, 4, 4) * 8) / 8)
