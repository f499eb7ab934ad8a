"""yolo355.tools.prepare: fp32 checkpoint -> fold -> quantize -> calibrate -> package."""
import numpy as np
import pytest
import torch

from oracle import yolo_oracle as O
from yolo355 import synth


def _fp32_sd(classes):
    layers = synth.make_fp32_model("slim_yolo_v2", 5, classes, 5, pred_gain=1.5, obj_bias=-2.0)
    return layers, {k: torch.from_numpy(v.copy()) for k, v in synth.state_dict_fp32(layers).items()}


def test_fold_uses_the_reference_formula():
    """CPU: fold_model == the oracle's restatement of utils/bn_fuse.py:21-45 (conv bias left unscaled)"""
    from yolo355.models import SlimYOLOv2
    from yolo355.tools.prepare import fold_model
    layers, sd = _fp32_sd(2)
    m = SlimYOLOv2("cpu", input_size=[96, 96], num_classes=2, anchor_size=synth.ANCHOR_SIZE_MASK)
    m.load_state_dict(sd, strict=False)
    m.eval()
    got = fold_model(m)
    for (w, b), L in zip(got, layers):
        if L["bn"] is None:
            assert np.array_equal(w.numpy(), L["w"]) and np.array_equal(b.numpy(), L["b"])
        else:
            rw, rb = O.fuse_conv_and_bn(L["w"], L["b"], *L["bn"])
            assert np.allclose(w.numpy(), rw, atol=1e-6, rtol=1e-6) and np.allclose(b.numpy(), rb, atol=1e-6, rtol=1e-6)


@pytest.mark.gpu
def test_prepare_end_to_end(tmp_path):
    from yolo355.tools import prepare as P
    H, W = 96, 160
    layers, sd = _fp32_sd(2)
    frames = synth.make_frames_u8(3, 2, H, W, "blocks")
    qm, package, report = P.prepare(sd, 2, synth.ANCHOR_SIZE_MASK, [H, W], frames)
    assert len(report) == 10 and all(r["fits_16bit"] == (r["headroom_bits"] > 0) for r in report)
    assert all(r["max_abs_output"] > 0 for r in report)
    # quantized weights == the oracle's recipe on the folded tensors
    folded = [(w.numpy(), b.numpy()) for w, b in P.fold_model(_model(sd, H, W))]
    ql = O.quantize_layers([("l%d" % i, w, b) for i, (w, b) in enumerate(folded)])
    for k, q in enumerate(ql):
        assert np.array_equal(package["q_w%d" % k], q["q_w"].astype(np.int8))
        assert np.array_equal(package["q_b%d" % k], q["q_b"].astype(np.int32))
        assert list(package["e%d" % k]) == [q["e_w"], q["e_b"]]
    # exponents == the oracle's first-call calibration on the same image
    x = synth.normalize_frames(frames)
    tr = [O.RangeTracker() for _ in range(11)]
    O.detect(x[:1], ql, tr, [H, W], synth.ANCHOR_SIZE_MASK, 2)
    assert [int(v) for v in package["sa"]] == [t.exponent() for t in tr]
    # the reported maxima are the ones that define the exponents
    for r, t in zip(report, tr[1:]):
        assert O.floor_log2_scale(np.float32(r["max_abs_output"]))[0] == t.exponent()
    # the package runs and equals the drop-in model
    path = str(tmp_path / "pkg.npz")
    np.savez_compressed(path, **package)
    eng = P.load_package(path, max_batch=2)
    a = eng.forward_frames(frames)
    b = qm.forward_frames(frames)
    for u, v in zip(a, b):
        for s, t in zip(u, v):
            assert np.array_equal(s, t)
    # the quantized checkpoint has the reference's key layout
    keys = set(qm.state_dict())
    assert {"conv1.convs.0.weight", "pred.bias", "a_tracker_in.scale", "a_tracker_pred.first_a"} <= keys
    eng.close()


def _model(sd, H, W):
    from yolo355.models import SlimYOLOv2
    m = SlimYOLOv2("cpu", input_size=[H, W], num_classes=2, anchor_size=synth.ANCHOR_SIZE_MASK)
    m.load_state_dict(sd, strict=False)
    m.eval()
    return m


@pytest.mark.gpu
def test_prepare_multi_batch_calibration():
    """--calib-batch N: the calibration loop of retune_bias_quantize.py:357-369 (EMA trackers over batches, stop after the
    batch in front of which more than `calib_images` images had been seen) == the oracle's trackers driven over the batch
    count the reference's own condition gives."""
    from yolo355.tools import prepare as P
    H, W = 96, 160
    layers, sd = _fp32_sd(2)
    frames = synth.make_frames_u8(7, 7, H, W, "blocks")
    qm, package, report = P.prepare(sd, 2, synth.ANCHOR_SIZE_MASK, [H, W], frames, calib_batch=2, calib_images=4)
    folded = [(w.numpy(), b.numpy()) for w, b in P.fold_model(_model(sd, H, W))]
    ql = O.quantize_layers([("l%d" % i, w, b) for i, (w, b) in enumerate(folded)])
    x = synth.normalize_frames(frames)
    tr = [O.RangeTracker() for _ in range(11)]
    assert P.calib_batches(7, 2, 4) == 4
    for i0 in (0, 2, 4, 6):          # reference condition `batch_size * iter_i > 4`, checked after batch iter_i (0-based): 4 batches
        O.forward_backbone_int(x[i0:i0 + 2], ql, tr, quant_freeze=False, saturate=True, keep=False)
    assert [int(v) for v in package["sa"]] == [t.exponent() for t in tr]
    got = [float(getattr(qm, n).scale.item()) for n in ("a_tracker_in", "a_tracker3_2", "a_tracker_pred")]
    ref = [float(tr[i].scale.item()) for i in (0, 4, 10)]
    assert np.allclose(got, ref, rtol=1e-6, atol=0)
    assert all(int(getattr(qm, "a_tracker%s" % s).first_a.item()) == 1 for s in ("_in", "1", "_pred"))


def test_calibration_loop_batch_count_is_the_references():
    """retune_bias_quantize.py:324,365-367: `for iter_i, batch in enumerate(loader): forward(batch); if batch_size * iter_i >
    1000: break` -- restated literally here and compared with tools.prepare.calib_batches (the loop used to stop
    one batch early)."""
    from yolo355.tools.prepare import calib_batches

    def reference_loop(n_images, bs, limit):
        done = 0
        for iter_i in range(-(-n_images // bs)):
            done += 1
            if bs * iter_i > limit:
                break
        return done

    assert calib_batches(5000, 32) == 33 == reference_loop(5000, 32, 1000)      # 1056 images
    for n, bs, lim in [(7, 2, 4), (5000, 16, 1000), (5000, 64, 1000), (100, 32, 1000), (1001, 1, 1000), (3000, 1000, 1000),
                       (10, 3, 0), (64, 64, 1000)]:
        assert calib_batches(n, bs, lim) == reference_loop(n, bs, lim), (n, bs, lim)
