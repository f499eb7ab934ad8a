"""Operator API of the wider model families: reorg_layer, SPP, 1x1 / stride-2 convolutions and
the residual add.  CPU: the numpy restatement against the reference's golden outputs.  GPU: the HIP operators
(through the C ABI and through the drop-in modules) against both."""
import os

import numpy as np
import pytest

from cases import OPS_CASES, ops_inputs
from oracle import ops_oracle as O

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ops.npz"))
# bf16 operands (rel. 2^-9 each) and a bf16 result (2^-9), K up to 64*9 products of O(1/sqrt(K)) terms:
# |err| <= 2^-8 |y| + 0.02 covers it with margin; the fixtures' outputs are O(1..4)
BF16_RTOL, BF16_ATOL = 2.0 ** -7, 0.03


def oracle_out(tag, kind, prm, d):
    if kind == "reorg":
        return O.reorg(d["x"], prm[4])
    if kind == "spp":
        return O.spp(d["x"])
    if kind == "conv":
        mod, B, Cin, Cout, H, W, k, s, leaky = prm
        slope = 0.1 if mod == "Conv_BN_LeakyReLU" else (0.125 if leaky else 0.0)
        return O.conv_bn_act(d["x"], d["w"], d["b"], d["bn_w"], d["bn_b"], d["bn_mean"], d["bn_var"], stride=s, neg_slope=slope)
    B, ch, H, W, nb = prm
    blocks = [tuple((d["w%d_%d" % (i, j)], d["b%d_%d" % (i, j)], tuple(d["bn%d_%d" % (i, j)])) for j in range(2)) for i in range(nb)]
    return O.resblock(d["x"], blocks)


@pytest.mark.parametrize("case", OPS_CASES, ids=[c[0] for c in OPS_CASES])
def test_oracle_matches_reference_golden(case):
    tag, kind, prm = case
    got = oracle_out(tag, kind, prm, ops_inputs(tag, kind, prm))
    ref = GOLD[tag]
    assert got.shape == ref.shape
    if kind in ("reorg", "spp"):
        assert np.array_equal(got.astype(np.float32), ref)          # data movement / max: bit-exact
    else:
        assert np.abs(got - ref).max() < 2e-5                       # fp32 reference vs float64 restatement


def _module_for(kind, prm, d):
    import torch
    from yolo355.utils import modules as M
    from yolo355.backbone import darknet as D

    def load(m, w, b, bn):
        m.convs[0].weight.data = torch.from_numpy(w.copy())
        m.convs[0].bias.data = torch.from_numpy(b.copy())
        m.convs[1].weight.data = torch.from_numpy(np.asarray(bn[0]).copy())
        m.convs[1].bias.data = torch.from_numpy(np.asarray(bn[1]).copy())
        m.convs[1].running_mean.data = torch.from_numpy(np.asarray(bn[2]).copy())
        m.convs[1].running_var.data = torch.from_numpy(np.asarray(bn[3]).copy())
    if kind == "reorg":
        return M.reorg_layer(prm[4])
    if kind == "spp":
        return M.SPP()
    if kind == "conv":
        mod, B, Cin, Cout, H, W, k, s, leaky = prm
        m = M.Conv2d(Cin, Cout, k, padding=k // 2, stride=s, leakyReLU=leaky) if mod == "Conv2d" \
            else D.Conv_BN_LeakyReLU(Cin, Cout, k, padding=k // 2, stride=s)
        load(m, d["w"], d["b"], [d["bn_w"], d["bn_b"], d["bn_mean"], d["bn_var"]])
        return m.eval()
    B, ch, H, W, nb = prm
    m = D.resblock(ch, nblocks=nb)
    for i in range(nb):
        for j in range(2):
            load(m.module_list[i][j], d["w%d_%d" % (i, j)], d["b%d_%d" % (i, j)], d["bn%d_%d" % (i, j)])
    return m.eval()


@pytest.mark.gpu
@pytest.mark.parametrize("case", OPS_CASES, ids=[c[0] for c in OPS_CASES])
def test_dropin_modules_match_reference(case):
    """the drop-in modules (same constructor arguments and state_dict layout as the reference's) on cuda:0"""
    import torch
    tag, kind, prm = case
    d = ops_inputs(tag, kind, prm)
    m = _module_for(kind, prm, d)
    xd = torch.from_numpy(d["x"]).cuda()
    with torch.no_grad():
        y = m(xd)
        # device-resident: a second call (weights already packed on the GPU) must not synchronise the host
        # from torch's side -- no .cpu() / .item() / host copy of a tensor inside forward
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            y2 = m(xd)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        assert torch.equal(y, y2)
        ycpu = m(torch.from_numpy(d["x"]))               # a CPU tensor takes the host-pointer entry points: same kernels, same bits
    assert y.is_cuda and y.dtype == torch.float32 and not ycpu.is_cuda
    got = y.cpu().numpy()
    assert np.array_equal(got, ycpu.numpy())
    ref = GOLD[tag]
    assert got.shape == ref.shape
    if kind in ("reorg", "spp"):
        assert np.array_equal(got, ref)
        return
    orc = oracle_out(tag, kind, prm, d)
    for want in (ref, orc):
        err = np.abs(got - want)
        tol = BF16_ATOL * (3 if kind == "resblock" else 1) + BF16_RTOL * np.abs(want)
        assert (err <= tol).all(), (tag, float(err.max()), float(np.abs(want).max()))
    # and the error is bf16-sized, not a layout bug hiding under the tolerance
    assert np.abs(got - ref).mean() < 0.01


@pytest.mark.gpu
def test_conv2d_bf16_is_exact_on_bf16_operands():
    """small integers are exact in bf16 and their sums exact in fp32: the MFMA path must reproduce them exactly
    (stride 2 on an odd map, 1x1, residual)"""
    from yolo355 import engine as E, synth
    rng = synth.uniform_pm1
    x = np.round(rng(1, (2, 48, 9, 11)) * 4).astype(np.float32)
    w = np.round(rng(2, (40, 48, 3, 3)) * 2).astype(np.float32)
    b = np.round(rng(3, (40,)) * 8).astype(np.float32)
    got = E.conv2d_bf16(x, w, b, stride=2, neg_slope=0.5)
    want = O.conv2d(x, w, b, stride=2)
    want = np.where(want >= 0, want, want * 0.5)
    assert np.abs(want).max() < 256                                   # representable in bf16 after the epilogue
    assert np.array_equal(got, want.astype(np.float32))
    w1 = np.round(rng(4, (64, 48, 1, 1)) * 2).astype(np.float32)
    res = np.round(rng(5, (2, 64, 9, 11)) * 16).astype(np.float32)
    got = E.conv2d_bf16(x, w1, None, residual=res, stride=1, neg_slope=1.0)
    want = O.conv2d(x, w1, np.zeros(64, np.float32)) + res
    assert np.abs(want).max() < 256
    assert np.array_equal(got, want.astype(np.float32))


@pytest.mark.gpu
def test_wider_ops_fail_loudly():
    from yolo355 import engine as E
    from yolo355._ffi import Y355Error
    x = np.zeros((1, 4, 7, 8), np.float32)
    with pytest.raises(Y355Error):
        E.reorg_f32(x, 2)                                             # 7 is not divisible by 2
    with pytest.raises(Y355Error):
        E.conv2d_bf16(x, np.zeros((8, 4, 1, 1), np.float32), stride=2)   # stride 2 needs a 3x3 kernel
    with pytest.raises(Y355Error):
        E.conv2d_bf16(x, np.zeros((8, 4, 5, 5), np.float32))


# ---- whole models of the wider families, composed from the operator API ---------------------------------
from cases import WIDE_MODEL_CASES, synth_state_dict  # noqa: E402
from helpers import dets_close  # noqa: E402

WGOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "models_wide.npz"))


@pytest.mark.gpu
@pytest.mark.parametrize("case", WIDE_MODEL_CASES, ids=[c[0] for c in WIDE_MODEL_CASES])
def test_yolo_v2_dropin_matches_reference(case):
    """myYOLOv2 (DarkNet-19, reorg route, stride-32 head) with the reference's constructor and state_dict layout:
    prediction map within the bf16 tolerance of the reference's fp32 map, detections close, and the head EXACT on
    the engine's own map (decode + NMS restated in numpy)."""
    import torch
    from yolo355 import synth, engine as E
    from yolo355.models.yolo_v2 import myYOLOv2
    from oracle import fp32_oracle as F
    tag, cls, size, classes, seed = case
    m = myYOLOv2("cuda", input_size=size, num_classes=classes, trainable=False, conf_thresh=0.05, nms_thresh=0.5,
                 anchor_size=synth.ANCHOR_SIZE)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed))
    m.eval()
    x = torch.from_numpy(synth.make_images(seed + 1, 1, size[0], size[1])).cuda()
    with torch.no_grad():
        pred = m.prediction_map(x)
    ref = WGOLD[tag + "_pred"]
    assert pred.shape == ref.shape
    err = np.abs(pred - ref)
    # 23 bf16 layers: the error of a map with |values| up to ~7 stays below 2 % of the range, mean below 0.5 %
    assert err.max() < 0.02 * np.abs(ref).max() + 0.05, float(err.max())
    assert err.mean() < 0.005 * np.abs(ref).max(), float(err.mean())
    b, s, c = m.forward_batch_composed(x)[0]
    assert b.dtype == np.float32 and s.dtype == np.float32 and c.dtype == np.int64 and b.flags.writeable
    fr, fg = dets_close((WGOLD[tag + "_boxes"], WGOLD[tag + "_scores"], WGOLD[tag + "_cls"]), (b, s, c), iou_min=0.7, score_tol=0.08)
    assert fr > 0.9 and fg > 0.9, (fr, fg)
    # the head itself: exact decode + NMS of the engine's own prediction map
    want = F.detect_v2(pred, synth.ANCHOR_SIZE, classes, size, 32, 0.05, 0.5)[0]
    assert len(want[1]) == len(s) and np.array_equal(want[2], c)
    assert np.abs(want[0] - b).max() < 2e-5 and np.abs(want[1] - s).max() < 2e-6
    # ---- the GPU-resident graph (y355_net, Y355_ARCH_YOLO_V2): same tolerances against the reference, and the
    # same numbers as the layer-by-layer form up to the bf16 rounding of intermediate maps (here: identical maps,
    # both forms round every activation to bf16 and accumulate in fp32 in the same order)
    b2, s2, c2 = m(x)
    assert b2.dtype == np.float32 and s2.dtype == np.float32 and c2.dtype == np.int64 and b2.flags.writeable
    fr, fg = dets_close((WGOLD[tag + "_boxes"], WGOLD[tag + "_scores"], WGOLD[tag + "_cls"]), (b2, s2, c2), iou_min=0.7, score_tol=0.08)
    assert fr > 0.9 and fg > 0.9, (fr, fg)
    net = m._get_net(1)
    pred2 = net.get_tensor(net.num_tensors - 1, 1)
    err = np.abs(pred2 - ref)
    assert err.max() < 0.02 * np.abs(ref).max() + 0.05 and err.mean() < 0.005 * np.abs(ref).max(), (float(err.max()), float(err.mean()))
    assert np.abs(pred2 - pred).max() < 0.02 * np.abs(ref).max()
    want = F.detect_v2(pred2, synth.ANCHOR_SIZE, classes, size, 32, 0.05, 0.5)[0]
    assert len(want[1]) == len(s2) and np.array_equal(want[2], c2)
    assert np.abs(want[0] - b2).max() < 2e-5 and np.abs(want[1] - s2).max() < 2e-6
    # batch semantics: element i of a batch equals the single-image run
    xb = torch.from_numpy(np.concatenate([synth.make_images(seed + 1 + i, 1, size[0], size[1]) for i in range(3)])).cuda()
    outs = m.forward_batch(xb)
    assert all(np.array_equal(p, q) for p, q in zip(outs[0], (b2, s2, c2)))


@pytest.mark.parametrize("case", WIDE_MODEL_CASES, ids=[c[0] for c in WIDE_MODEL_CASES])
def test_v2_head_oracle_matches_reference_golden(case):
    """CPU: decode + per-class NMS of the reference's own prediction map reproduce the reference's detections"""
    from yolo355 import synth
    from oracle import fp32_oracle as F
    tag, cls, size, classes, seed = case
    d = F.detect_v2(WGOLD[tag + "_pred"], synth.ANCHOR_SIZE, classes, size, 32, 0.05, 0.5)[0]
    assert np.array_equal(d[0], WGOLD[tag + "_boxes"]) and np.array_equal(d[1], WGOLD[tag + "_scores"])
    assert np.array_equal(d[2], WGOLD[tag + "_cls"])


from cases import WIDE3_MODEL_CASES  # noqa: E402


def _gold3(tag):
    return [WGOLD[tag + "_pred_%d" % k].astype(np.float32) for k in (1, 2, 3)]


@pytest.mark.parametrize("case", WIDE3_MODEL_CASES, ids=[c[0] for c in WIDE3_MODEL_CASES])
def test_v3_head_oracle_matches_reference_golden(case):
    """CPU: the three-level decode + NMS restatement on the reference's own maps reproduces its detections
    (the stride-8 map is stored as float16, so: same count and classes, boxes / scores within the rounding)"""
    from yolo355 import synth
    from oracle import fp32_oracle as F
    tag = case[0]
    d = F.detect_v3(_gold3(tag), synth.MULTI_ANCHOR_SIZE, case[4], case[3], 0.05, 0.5)[0]
    ok = dets_close((WGOLD[tag + "_boxes"], WGOLD[tag + "_scores"], WGOLD[tag + "_cls"]), d, iou_min=0.95, score_tol=0.005)
    assert min(ok) > 0.97, ok
    assert abs(len(d[1]) - len(WGOLD[tag + "_scores"])) <= max(2, len(d[1]) // 50)


@pytest.mark.gpu
@pytest.mark.parametrize("case", WIDE3_MODEL_CASES, ids=[c[0] for c in WIDE3_MODEL_CASES])
def test_yolo_v3_dropins_match_reference(case):
    """myYOLOv3 / myYOLOv3Spp (DarkNet-53: stride-2 convolutions, 23 residual blocks; 1x1 + bilinear x2 routes; SPP)
    composed from the operator API: the three prediction maps within the bf16 tolerance of the reference's fp32
    maps, detections close, head exact on the engine's own maps."""
    import torch
    from yolo355 import synth
    from yolo355.models import yolo_v3 as Y
    from oracle import fp32_oracle as F
    tag, _, cls, size, classes, seed, gain = case
    m = getattr(Y, cls)("cuda", input_size=size, num_classes=classes, trainable=False, conf_thresh=0.05, nms_thresh=0.5,
                        anchor_size=synth.MULTI_ANCHOR_SIZE)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed, weight_gain=gain))
    m.eval()
    x = torch.from_numpy(synth.make_images(seed + 1, 1, size[0], size[1])).cuda()
    with torch.no_grad():
        preds = m.prediction_maps(x)
    for got, ref in zip(preds, _gold3(tag)):
        assert got.shape == ref.shape
        err = np.abs(got - ref)
        # 75+ bf16 layers with residual accumulation: below 3 % of the range, mean below 0.6 %
        assert err.max() < 0.03 * np.abs(ref).max() + 0.03, float(err.max())
        assert err.mean() < 0.006 * np.abs(ref).max(), float(err.mean())
    b, s, c = m.forward_batch_composed(x)[0]
    fr, fg = dets_close((WGOLD[tag + "_boxes"], WGOLD[tag + "_scores"], WGOLD[tag + "_cls"]), (b, s, c), iou_min=0.7, score_tol=0.05)
    assert fr > 0.9 and fg > 0.9, (fr, fg, len(s), len(WGOLD[tag + "_scores"]))
    want = F.detect_v3(preds, synth.MULTI_ANCHOR_SIZE, classes, size, 0.05, 0.5)[0]
    assert len(want[1]) == len(s) and np.array_equal(want[2], c)
    assert np.abs(want[0] - b).max() < 2e-5 and np.abs(want[1] - s).max() < 2e-6
    # ---- the GPU-resident graph (y355_net, Y355_ARCH_YOLO_V3 / _SPP)
    b2, s2, c2 = m(x)
    net = m._get_net(1)
    nt = net.num_tensors
    preds2 = [net.get_tensor(nt - 1, 1), net.get_tensor(nt - 3, 1), net.get_tensor(nt - 5, 1)]     # strides 8, 16, 32
    for got, ref in zip(preds2, _gold3(tag)):
        assert got.shape == ref.shape
        err = np.abs(got - ref)
        assert err.max() < 0.03 * np.abs(ref).max() + 0.03 and err.mean() < 0.006 * np.abs(ref).max(), (float(err.max()), float(err.mean()))
    fr, fg = dets_close((WGOLD[tag + "_boxes"], WGOLD[tag + "_scores"], WGOLD[tag + "_cls"]), (b2, s2, c2), iou_min=0.7, score_tol=0.05)
    assert fr > 0.9 and fg > 0.9, (fr, fg, len(s2))
    want = F.detect_v3(preds2, synth.MULTI_ANCHOR_SIZE, classes, size, 0.05, 0.5)[0]
    assert len(want[1]) == len(s2) and np.array_equal(want[2], c2)
    assert np.abs(want[0] - b2).max() < 2e-5 and np.abs(want[1] - s2).max() < 2e-6


@pytest.mark.gpu
def test_yolo_v3_at_416_runs_on_the_gpu_resident_graph():
    """the reference's default input size: 10 647 anchors per image -> threshold-then-compact head; batch of two; the head
    is exact on the engine's own maps, and element i of the batch equals the single-image run"""
    import torch
    from yolo355 import synth
    from yolo355.models.yolo_v3 import myYOLOv3
    from oracle import fp32_oracle as F
    size, classes, seed = [416, 416], 20, 4200
    m = myYOLOv3("cuda", input_size=size, num_classes=classes, trainable=False, conf_thresh=0.05, nms_thresh=0.5,
                 anchor_size=synth.MULTI_ANCHOR_SIZE)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed, weight_gain=1.3))
    m.eval()
    x = torch.from_numpy(np.concatenate([synth.make_images(seed + 1 + i, 1, 416, 416) for i in range(2)])).cuda()
    outs = m.forward_batch(x)
    net = m._get_net(2)
    assert net.num_anchors_total == 10647
    nt = net.num_tensors
    preds = [net.get_tensor(nt - 1, 2), net.get_tensor(nt - 3, 2), net.get_tensor(nt - 5, 2)]
    want = F.detect_v3(preds, synth.MULTI_ANCHOR_SIZE, classes, size, 0.05, 0.5)
    for (b, s, c), w in zip(outs, want):
        assert len(s) > 20 and len(w[1]) == len(s) and np.array_equal(w[2], c)
        assert np.abs(w[0] - b).max() < 2e-5 and np.abs(w[1] - s).max() < 2e-6
    one = m.forward_batch(x[1:2])[0]
    assert all(np.array_equal(p, q) for p, q in zip(one, outs[1]))


@pytest.mark.gpu
def test_head_with_more_than_4096_anchors():
    """yolo_v3 at 416 x 416 has 10 647 anchors per image: threshold-then-compact in front of the sort.  Exact against
    the numpy restatement on the same maps; loud when more than 4096 anchors pass the threshold."""
    from yolo355 import engine as E, synth
    from yolo355._ffi import Y355Error
    from oracle import fp32_oracle as F
    C, A, size = 20, 3, [416, 416]
    preds = []
    for li, s in enumerate((8, 16, 32)):
        hs = 416 // s
        p = synth.uniform_pm1(9100 + li, (2, A * (5 + C), hs, hs)).astype(np.float32) * 3.0
        p[:, :A] -= 3.5                                              # objectness: a minority of the anchors pass 0.05
        preds.append(p)
    got = E.head_f32(preds, (8, 16, 32), np.asarray(synth.MULTI_ANCHOR_SIZE, np.float32).reshape(3, A, 2), C, size, 1.0, 0.05, 0.5)
    want = F.detect_v3(preds, synth.MULTI_ANCHOR_SIZE, C, size, 0.05, 0.5)
    for g, w in zip(got, want):
        assert 100 < len(w[1]) and len(g[1]) == len(w[1]), (len(g[1]), len(w[1]))
        assert np.array_equal(g[2], w[2])
        assert np.abs(g[0] - w[0]).max() < 2e-5 and np.abs(g[1] - w[1]).max() < 2e-6
    with pytest.raises(Y355Error, match="4096"):
        E.head_f32(preds, (8, 16, 32), np.asarray(synth.MULTI_ANCHOR_SIZE, np.float32).reshape(3, A, 2), C, size, 1.0, 1e-6, 0.5)


def test_wider_models_keep_the_reference_state_dict_layout():
    """CPU: key order and shapes of myYOLOv2 / myYOLOv3 / myYOLOv3Spp equal the reference classes' (recorded by
    tests/golden/gen_golden_models_wide.py), and the weight-slot lists of the GPU graphs cover every convolution once"""
    import json
    from yolo355 import synth
    from yolo355.models.yolo_v2 import myYOLOv2
    from yolo355.models.yolo_v3 import myYOLOv3, myYOLOv3Spp
    lay = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "models_wide_layout.json")))
    for cls, anchors, nconv in ((myYOLOv2, synth.ANCHOR_SIZE, 23), (myYOLOv3, synth.MULTI_ANCHOR_SIZE, 75), (myYOLOv3Spp, synth.MULTI_ANCHOR_SIZE, 75)):
        m = cls("cpu", input_size=[224, 224], num_classes=20, trainable=False, anchor_size=anchors)
        got = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        assert got == lay[cls.__name__]
        mods = m._conv_modules()
        convs = [mm[0] if hasattr(mm, "__getitem__") else mm for mm in mods]
        assert len(convs) == nconv and len({id(c) for c in convs}) == nconv
        import torch.nn as nn
        assert sum(isinstance(x, nn.Conv2d) for x in m.modules()) == nconv


@pytest.mark.gpu
def test_device_resident_operators_equal_the_host_pointer_forms():
    """y355_*_dev / y355_conv_op against the host-pointer entry points they shadow: same kernels, identical bits; ConvOp re-used
    across batch sizes and map sizes (workspaces grow, halos re-zeroed), the int8 form on dyadic and on non-dyadic inputs."""
    import torch
    from yolo355 import engine as E, synth
    dev = torch.device("cuda", 0)
    rng = synth.uniform_pm1
    x = rng(1, (3, 24, 20, 28)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev)
    assert np.array_equal(E.reorg_f32_dev(xd, 2).cpu().numpy(), E.reorg_f32(x, 2))
    assert np.array_equal(E.spp_f32_dev(xd).cpu().numpy(), E.spp_f32(x))
    assert np.array_equal(E.maxpool2x2_f32_dev(xd).cpu().numpy(), E.maxpool2x2_f32(x))
    assert np.array_equal(E.upsample2x_f32_dev(xd).cpu().numpy(), E.upsample2x_f32(x))
    # bf16 convolutions: 3x3, 3x3 stride 2 with a residual, 1x1 with fp32 output; one operator object, several geometries
    for k, s, res, f32 in ((3, 1, False, False), (3, 2, True, False), (1, 1, False, True)):
        w = rng(10 + k + s, (40, 24, k, k)).astype(np.float32) * 0.2
        b = rng(20 + k, (40,)).astype(np.float32)
        op = E.ConvOp.bf16(w, b, stride=s, neg_slope=0.1, device=dev)
        for shape in ((3, 24, 20, 28), (1, 24, 9, 11), (5, 24, 20, 28), (2, 24, 33, 17)):
            xs = rng(30 + shape[0] + shape[2], shape).astype(np.float32)
            Ho, Wo = ((shape[2] + 1) // 2, (shape[3] + 1) // 2) if s == 2 else shape[2:]
            r = rng(40 + shape[2], (shape[0], 40, Ho, Wo)).astype(np.float32) if res else None
            want = E.conv2d_bf16(xs, w, b, residual=r, stride=s, neg_slope=0.1, out_fp32=f32)
            got = op.forward(torch.from_numpy(xs).to(dev), None if r is None else torch.from_numpy(r).to(dev), out_fp32=f32)
            assert np.array_equal(got.cpu().numpy(), want), (k, s, shape)
        op.close()
    # int8: Conv2d_fuse on dyadic operands
    qw = (synth.uniform_u8(5, (48, 32, 3, 3)).astype(np.int32) - 128).clip(-127, 127)
    qb = (synth.uniform_u8(6, (48,)).astype(np.int32) - 128).clip(-127, 127)
    op = E.ConvOp.int8(qw, qb, 9, 6, leaky=True, device=dev)
    for shape, sa in (((2, 32, 16, 24), 4), ((1, 32, 7, 9), 6), ((3, 32, 16, 24), 2)):
        q = (synth.uniform_u8(7 + sa, shape).astype(np.int32) - 128).clip(-127, 127)
        q.flat[0] = 127                                    # the tensor's exponent is then exactly sa
        xq = (q.astype(np.float32) * np.float32(2.0 ** -sa))
        t, frac = E.conv3x3_i8_raw(q, qw, qb, sa, 9, 6, leaky=True)
        want = t.astype(np.float32) * np.float32(2.0 ** -frac)
        got = op.forward_i8(torch.from_numpy(xq).to(dev))
        assert got is not None and np.array_equal(got.cpu().numpy(), want), shape
    assert op.forward_i8(torch.from_numpy(rng(9, (2, 32, 16, 24)).astype(np.float32)).to(dev)) is None      # not dyadic: the caller's bf16 route
    assert op.forward_i8(torch.zeros((1, 32, 8, 8), device=dev)) is None
    op.close()
