#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference, read-only).  Nothing of the
reference's source travels: this script imports it, feeds it inputs/weights from the
build-owned generator (yolo355.synth) and stores inputs' seeds + the reference's outputs
as small .npz fixtures.  Import recipe: SURVEY.md section 8(c).

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.npz
"""
import importlib
import os
import sys
import types
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "yolo-compression-and-deployment-in-fpga_amd"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    _stub("cv2")
    _stub("pycocotools")
    _stub("pycocotools.coco", COCO=object)
    _stub("pycocotools.cocoeval", COCOeval=object)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    np.int = int
    np.bool = bool
    sys.path.insert(0, REF)
    ref = types.SimpleNamespace()
    m = importlib.import_module("models.slim_yolo_v2")
    ref.SlimYOLOv2 = m.SlimYOLOv2
    ref.Q = m.SlimYOLOv2_quantize_bnfuse
    ref.Tracker = m.AveragedRangeTracker
    ref.rbq = importlib.import_module("retune_bias_quantize")
    ref.rbqf = importlib.import_module("retune_bias_quantize_findbest")
    ref.fuse_conv_and_bn = importlib.import_module("utils.bn_fuse").fuse_conv_and_bn
    ref.modules = importlib.import_module("utils.modules")
    ref.config = importlib.import_module("data.config")
    return ref


import torch  # noqa: E402
from yolo355 import synth  # noqa: E402

KEYS = ["conv1", "conv2", "conv3_1", "conv3_2", "conv4_1", "conv4_2", "conv5", "conv6", "conv7"]


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def load_weights(model, weights):
    sd = model.state_dict()
    for name, w, b in weights:
        if name == "pred":
            sd["pred.weight"] = torch.from_numpy(w.copy())
            sd["pred.bias"] = torch.from_numpy(b.copy())
        else:
            sd[name + ".convs.0.weight"] = torch.from_numpy(w.copy())
            sd[name + ".convs.0.bias"] = torch.from_numpy(b.copy())
    model.load_state_dict(sd)


def build_q_model(ref, weights, num_classes, anchors, size, conf, quantizer):
    model = ref.Q("cpu", input_size=size, num_classes=num_classes, trainable=False,
                  conf_thresh=conf, nms_thresh=0.5, anchor_size=anchors)
    load_weights(model, weights)
    quantizer.quantized_layers.clear()
    quantizer.init_quantize_net(model, 8)
    quantizer.quantize_layers(8)
    model.eval()
    return model


class Recorder:
    """Wraps AveragedRangeTracker.quantize_activation to tap the fake-quant outputs."""

    def __init__(self, ref):
        self.ref = ref
        self.orig = ref.Tracker.quantize_activation
        self.taps = []

    def __enter__(self):
        rec = self

        def wrapped(self_t, activation, *a, **k):
            out = rec.orig(self_t, activation, *a, **k)
            e = int(torch.floor(torch.log2(self_t.scale)).item())
            q = torch.round(out.detach() * (2.0 ** e))
            rec.taps.append((e, q.to(torch.int32).numpy(), float(activation.abs().max())))
            return out
        self.ref.Tracker.quantize_activation = wrapped
        return self

    def __exit__(self, *a):
        self.ref.Tracker.quantize_activation = self.orig


def pool_np(q):
    b, c, h, w = q.shape
    return q.reshape(b, c, h // 2, 2, w // 2, 2).max(axis=(3, 5))


def run_e2e(ref, tag, wkw, size, num_classes, anchors, calib_seed, img_seeds, confs, out,
            quantizer=None, find=False, pattern="noise"):
    """End-to-end fixture: calibrate on image `calib_seed` (first call), then run the
    frozen model on each of img_seeds, one image at a time as the reference does."""
    quantizer = quantizer or ref.rbq
    weights = synth.make_weights(**wkw, num_classes=num_classes)
    model = build_q_model(ref, weights, num_classes, anchors, size, confs[0], quantizer)
    xc = synth.make_images(calib_seed, 1, size[0], size[1], pattern)
    with Recorder(ref) as rec, torch.no_grad():
        d0 = model(torch.from_numpy(xc), quantization=True, find=find)
    sa = [t[0] for t in rec.taps]
    out[tag + "/sa"] = np.array(sa, np.int32)
    pools = [False, True, True, False, True, False, True, False, False, False, False]
    for li, (e, q, amax) in enumerate(rec.taps):
        qq = pool_np(q) if pools[li] else q
        out[tag + "/calib/map_crc/%d" % li] = np.array([crc(qq.astype(np.int8)), int(np.abs(q).max()),
                                                         int(qq.astype(np.int64).sum() & 0x7FFFFFFF)], np.int64)
    out[tag + "/calib/pred_q"] = rec.taps[10][1].astype(np.int8)
    out[tag + "/calib/absmax"] = np.array([t[2] for t in rec.taps], np.float32)
    for ci, conf in enumerate(confs):
        model.conf_thresh = conf
        with torch.no_grad():
            b, s, c = model(torch.from_numpy(xc), quantization=True, find=find)
        out[tag + "/calib/det%d/boxes" % ci] = b.astype(np.float32)
        out[tag + "/calib/det%d/scores" % ci] = s.astype(np.float32)
        out[tag + "/calib/det%d/cls" % ci] = c.astype(np.int64)
    model.conf_thresh = confs[0]
    for si, seed in enumerate(img_seeds):
        x = synth.make_images(seed, 1, size[0], size[1], pattern)
        with Recorder(ref) as rec, torch.no_grad():
            b, s, c = model(torch.from_numpy(x), quantization=True, find=find)
        out[tag + "/img%d/boxes" % si] = b.astype(np.float32)
        out[tag + "/img%d/scores" % si] = s.astype(np.float32)
        out[tag + "/img%d/cls" % si] = c.astype(np.int64)
        out[tag + "/img%d/pred_q" % si] = np.clip(rec.taps[10][1], -128, 127).astype(np.int8)
        out[tag + "/img%d/qmax" % si] = np.array([int(np.abs(t[1]).max()) for t in rec.taps], np.int32)
        out[tag + "/img%d/nover" % si] = np.array([int((np.abs(t[1]) > 127).sum()) for t in rec.taps], np.int64)
    out[tag + "/meta"] = np.array([size[0], size[1], num_classes, calib_seed] + list(img_seeds), np.int64)
    out[tag + "/confs"] = np.array(confs, np.float32)
    return model


def gen_weight_prep(ref, out):
    """G2: reference quantize_tensor / quantize_tensor_b on the synthetic weights, and
    fuse_conv_and_bn on synthetic BN statistics."""
    for tag, kw in [("w2", dict(seed=2)), ("w3gap", dict(seed=3, bias_gain=40.0, weight_gain=3.0))]:
        ws = synth.make_weights(**kw, num_classes=2)
        for li, (name, w, b) in enumerate(ws):
            qw, sw = ref.rbq.quantize_tensor(torch.from_numpy(w), 8, False)
            qb, sb = ref.rbq.quantize_tensor_b(torch.from_numpy(b), 8, False)
            out["prep/%s/%d" % (tag, li)] = np.array(
                [int(np.log2(float(sw.reshape(-1)[0]))), int(np.log2(float(sb.reshape(-1)[0]))),
                 crc(qw.numpy().astype(np.int8)), crc(qb.numpy().astype(np.int8)),
                 int(qw.abs().max()), int(qb.abs().max())], np.int64)
    # BN fold, with and without a conv bias (the reference formula is only exact without)
    for case, (cin, cout, with_bias) in enumerate([(3, 16, True), (16, 32, False), (64, 64, True)]):
        w = synth.uniform_pm1(100 + case, (cout, cin, 3, 3)) * np.float32(0.2)
        b = synth.uniform_pm1(200 + case, (cout,)) * np.float32(0.3)
        g, be, mu, var = synth.make_bn(300 + case, cout)
        conv = torch.nn.Conv2d(cin, cout, 3, 1, 1, bias=with_bias)
        bn = torch.nn.BatchNorm2d(cout)
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(w))
            if with_bias:
                conv.bias.copy_(torch.from_numpy(b))
            bn.weight.copy_(torch.from_numpy(g)); bn.bias.copy_(torch.from_numpy(be))
            bn.running_mean.copy_(torch.from_numpy(mu)); bn.running_var.copy_(torch.from_numpy(var))
        fused = ref.fuse_conv_and_bn(conv, bn)
        out["fuse/%d/w" % case] = fused.weight.detach().numpy()
        out["fuse/%d/b" % case] = fused.bias.detach().numpy()
        out["fuse/%d/meta" % case] = np.array([cin, cout, int(with_bias)], np.int64)


def gen_layer_cases(ref, out):
    """G1: single fused layer through the reference's own Conv2d_fuse + tracker."""
    cases = [
        # cin, cout, h, w, sa_in, e_w, e_b, sa_out, leaky
        (3, 16, 8, 8, 4, 9, 9, 5, True),
        (16, 32, 12, 10, 5, 8, 6, 6, True),       # e_b < sa_in+e_w : bias left shift
        (64, 64, 26, 26, 6, 9, 5, 7, True),
        (32, 16, 6, 6, 2, 3, 9, -3, True),         # e_b > sa_in+e_w : accumulator left shift
        (256, 35, 13, 13, 7, 9, 10, 4, False),     # pred-like, no activation
        (128, 125, 4, 6, 6, 10, 5, 3, False),
    ]
    for ci, (cin, cout, h, w, sa_in, e_w, e_b, sa_out, leaky) in enumerate(cases):
        q_in = (synth.uniform_u8(500 + ci, (2, cin, h, w)).astype(np.int32) - 128).clip(-127, 127)
        q_w = (synth.uniform_u8(600 + ci, (cout, cin, 3, 3)).astype(np.int32) - 128).clip(-127, 127)
        q_b = (synth.uniform_u8(700 + ci, (cout,)).astype(np.int32) - 128).clip(-127, 127)
        if leaky:
            mod = ref.modules.Conv2d_fuse(cin, cout, 3, 1, leakyReLU=True)
            conv = mod.convs[0]
        else:
            mod = torch.nn.Conv2d(cin, cout, 3, 1, padding=1)
            conv = mod
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(q_w.astype(np.float32) / np.float32(2.0 ** e_w)))
            conv.bias.copy_(torch.from_numpy(q_b.astype(np.float32) / np.float32(2.0 ** e_b)))
        tr = ref.Tracker()
        tr.scale.fill_(float(2.0 ** sa_out) * 1.3)   # floor(log2(.)) == sa_out
        tr.first_a.fill_(1)
        x = torch.from_numpy(q_in.astype(np.float32) / np.float32(2.0 ** sa_in))
        with torch.no_grad():
            y = mod(x)
            yq = tr.quantize_activation(y, 8, True, True, True)
        q_out = torch.round(yq * (2.0 ** sa_out)).to(torch.int32).numpy()
        out["layer/%d/meta" % ci] = np.array([cin, cout, h, w, sa_in, e_w, e_b, sa_out, int(leaky),
                                               500 + ci, 600 + ci, 700 + ci], np.int64)
        out["layer/%d/q_out" % ci] = q_out.astype(np.int32)
        out["layer/%d/ymax" % ci] = np.array([float(y.abs().max())], np.float32)


def main():
    ref = import_reference()
    mask = ref.config.ANCHOR_SIZE_MASK
    assert mask == synth.ANCHOR_SIZE_MASK and ref.config.ANCHOR_SIZE == synth.ANCHOR_SIZE
    small = {}
    gen_weight_prep(ref, small)
    gen_layer_cases(ref, small)
    np.savez_compressed(os.path.join(HERE, "prep_layers.npz"), **small)

    e2e = {}
    # G3: the C1 / C2 configuration: 416x416, 2 classes, mask anchors, conf 0.01 and 0.1
    run_e2e(ref, "c1", dict(seed=2), [416, 416], 2, mask, 1, [0], [0.01, 0.1], e2e)
    # G4: sparse detections (objectness bias pushed negative)
    run_e2e(ref, "sparse", dict(seed=2, pred_gain=400.0, obj_bias=-4.0), [416, 416], 2, mask, 1, [0],
            [0.01, 0.1], e2e)
    # diverse scores (20 classes, structured image): no score ties -> strict NMS equality
    run_e2e(ref, "diverse", dict(seed=2, pred_gain=400.0, obj_bias=-4.0), [416, 416], 20,
            ref.config.ANCHOR_SIZE, 1, [0], [0.01, 0.1], e2e, pattern="blocks")
    # G5: the retune / find=True path must reproduce G3 exactly
    run_e2e(ref, "find", dict(seed=2), [416, 416], 2, mask, 1, [0], [0.01, 0.1], e2e,
            quantizer=ref.rbqf, find=True)
    # exponent-gap weights (bias shifts both ways), 20 classes, VOC anchors, non-square input
    run_e2e(ref, "gap", dict(seed=3, bias_gain=40.0, weight_gain=3.0), [320, 416], 20,
            ref.config.ANCHOR_SIZE, 11, [12], [0.01, 0.3], e2e)
    # G7: batch semantics, FPGA demo size 240x320: 4 images through one frozen model
    run_e2e(ref, "batch", dict(seed=2), [240, 320], 2, mask, 21, [22, 23, 24, 25], [0.01], e2e)
    np.savez_compressed(os.path.join(HERE, "e2e.npz"), **e2e)

    # G5b: the 2^15 head-room guard trips (slim_yolo_v2.py:222-227)
    guard = {}
    ws = dict(seed=2, weight_gain=6.0)
    model = build_q_model(ref, synth.make_weights(**ws, num_classes=2), 2, mask, [96, 96], 0.01, ref.rbqf)
    x = synth.make_images(1, 1, 96, 96)
    tripped = 0
    try:
        with torch.no_grad():
            model(torch.from_numpy(x), quantization=True, find=True)
    except AssertionError:
        tripped = 1
    guard["guard/tripped"] = np.array([tripped], np.int64)
    guard["guard/meta"] = np.array([96, 96, 2, 1, 6], np.int64)
    np.savez_compressed(os.path.join(HERE, "guard.npz"), **guard)
    for f in ("prep_layers.npz", "e2e.npz", "guard.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
