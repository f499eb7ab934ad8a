#!/usr/bin/env python3
"""Round-2 golden vectors (tests/golden/r2.npz), produced by running the REFERENCE in the build container
(import recipe and helpers of gen_golden.py; nothing of the reference's source is stored -- seeds and outputs only).

  qf32/*   SlimYOLOv2_quantize_bnfuse called the way test.py:84 / demo.py:81 / utils/vocapi_evaluator.py:67 call it:
           net(x) with the default quantization=False on a checkpoint whose weights are already power-of-two
           quantized (models/slim_yolo_v2.py:212-358 with the trackers as identity): pred map + detections.
  relu/*   Conv2d_fuse / Conv2d_fuse_nobias with leakyReLU=False (ReLU, utils/modules.py:26,37) + tracker.
  bt/*     the reference's own BaseTransform (data/__init__.py:30-56) followed by test.py:79-80 (BGR->RGB,
           HWC->CHW) on synthetic uint8 frames; cv2.resize is stubbed with the identity because the frames
           are generated at the network size (cv2 is not installed here; the resize itself is parity-unpinned).
  ema/*    AveragedRangeTracker in its non-frozen (EMA) branch (:30-31): the q_bf model with trainable=True run over
           five batches (a) as retune_bias_quantize.py:357-369 does (first batch on the un-quantized weights,
           weights quantized after every batch) and (b) with the weights quantized before the first batch, which
           is the order the engine's calibration loop can reproduce; exponents after every batch.

    python tests/golden/gen_golden_r2.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

import torch  # noqa: E402
from yolo355 import synth  # noqa: E402

QF32_CASES = [
    # tag, weights kwargs, size, classes, anchors name, image seeds, pattern
    ("qf_416", dict(seed=2, weight_gain=2.2, pred_gain=1.5, obj_bias=-2.0), [416, 416], 2, "mask", [0], "blocks"),
    ("qf_b2", dict(seed=5, weight_gain=2.2, pred_gain=1.5, obj_bias=-2.0), [240, 320], 2, "mask", [22, 23], "noise"),
    ("qf_voc", dict(seed=6, weight_gain=2.2, pred_gain=1.5, obj_bias=-2.0), [96, 160], 20, "voc", [7], "blocks"),
]
RELU_CASES = [
    # cin, cout, h, w, sa_in, e_w, e_b, sa_out, nobias
    (16, 24, 10, 12, 5, 8, 6, 6, False),
    (64, 32, 13, 13, 6, 9, 5, 7, False),
    (32, 48, 6, 9, 4, 7, 0, 5, True),
]
EMA = dict(weights=dict(seed=2), size=[96, 160], classes=2, batch=2, seeds=[41, 42, 43, 44, 45])


def gen_qf32(ref, out):
    for tag, wkw, size, classes, an, seeds, pattern in QF32_CASES:
        anchors = ref.config.ANCHOR_SIZE_MASK if an == "mask" else ref.config.ANCHOR_SIZE
        weights = synth.make_weights(**wkw, num_classes=classes)
        model = G.build_q_model(ref, weights, classes, anchors, size, 0.01, ref.rbq)   # dyadic weights in the modules
        grabbed = {}
        hook = model.pred.register_forward_hook(lambda m, i, o: grabbed.__setitem__("pred", o.detach().numpy().copy()))
        preds = []
        for bi, seed in enumerate(seeds):
            x = synth.make_images(seed, 1, size[0], size[1], pattern)
            for conf in (0.01, 0.1):
                model.conf_thresh = conf
                with torch.no_grad():
                    b, s, c = model(torch.from_numpy(x))                     # the canonical call: quantization=False
                out["%s/%d/det%g/boxes" % (tag, bi, conf)] = b.astype(np.float32)
                out["%s/%d/det%g/scores" % (tag, bi, conf)] = s.astype(np.float32)
                out["%s/%d/det%g/cls" % (tag, bi, conf)] = c.astype(np.int64)
            preds.append(grabbed["pred"])
        hook.remove()
        out[tag + "/pred"] = np.concatenate(preds).astype(np.float32)
        out[tag + "/meta"] = np.array([size[0], size[1], classes] + list(seeds), np.int64)
        print(tag, out[tag + "/pred"].shape, "max|pred| %.3f" % np.abs(out[tag + "/pred"]).max(),
              [len(out["%s/%d/det0.01/scores" % (tag, i)]) for i in range(len(seeds))],
              [len(out["%s/%d/det0.1/scores" % (tag, i)]) for i in range(len(seeds))])


def gen_relu(ref, out):
    for ci, (cin, cout, h, w, sa_in, e_w, e_b, sa_out, nobias) in enumerate(RELU_CASES):
        q_in = (synth.uniform_u8(900 + ci, (2, cin, h, w)).astype(np.int32) - 128).clip(-127, 127)
        q_w = (synth.uniform_u8(910 + ci, (cout, cin, 3, 3)).astype(np.int32) - 128).clip(-127, 127)
        q_b = (synth.uniform_u8(920 + ci, (cout,)).astype(np.int32) - 128).clip(-127, 127)
        cls = ref.modules.Conv2d_fuse_nobias if nobias else ref.modules.Conv2d_fuse
        mod = cls(cin, cout, 3, 1, leakyReLU=False)
        assert isinstance(mod.convs[1], torch.nn.ReLU)
        with torch.no_grad():
            mod.convs[0].weight.copy_(torch.from_numpy(q_w.astype(np.float32) / np.float32(2.0 ** e_w)))
            if not nobias:
                mod.convs[0].bias.copy_(torch.from_numpy(q_b.astype(np.float32) / np.float32(2.0 ** e_b)))
        tr = ref.Tracker()
        tr.scale.fill_(float(2.0 ** sa_out) * 1.3)
        tr.first_a.fill_(1)
        x = torch.from_numpy(q_in.astype(np.float32) / np.float32(2.0 ** sa_in))
        with torch.no_grad():
            y = mod(x)
            yq = tr.quantize_activation(y, 8, True, True, True)
        out["relu/%d/meta" % ci] = np.array([cin, cout, h, w, sa_in, e_w, e_b, sa_out, int(nobias), 900 + ci, 910 + ci, 920 + ci], np.int64)
        out["relu/%d/q_out" % ci] = torch.round(yq * (2.0 ** sa_out)).to(torch.int32).numpy()
        out["relu/%d/y" % ci] = y.numpy().astype(np.float32)


def gen_base_transform(ref, out):
    import cv2                                    # the stub installed by import_reference
    cv2.resize = lambda image, dsize: image       # frames are generated at the network size
    data = sys.modules["data"]
    for ci, (h, w, seed, pattern) in enumerate([(32, 48, 41, "noise"), (416, 416, 42, "blocks"), (240, 320, 43, "noise")]):
        frame = synth.make_frames_u8(seed, 1, h, w, pattern)[0]
        img, _, _ = data.BaseTransform([h, w])(frame)
        x = torch.from_numpy(img[:, :, (2, 1, 0)]).permute(2, 0, 1).unsqueeze(0).numpy()     # test.py:79-80
        x = np.ascontiguousarray(x, dtype=np.float32)
        out["bt/%d/meta" % ci] = np.array([h, w, seed, {"noise": 0, "blocks": 1}[pattern]], np.int64)
        out["bt/%d/crc" % ci] = np.array([G.crc(x)], np.int64)
        if h * w <= 4096:
            out["bt/%d/x" % ci] = x
        print("bt", ci, x.shape, x.dtype, "crc %08x" % G.crc(x))


def gen_ema(ref, out):
    cfg = EMA
    size, classes = cfg["size"], cfg["classes"]
    anchors = ref.config.ANCHOR_SIZE_MASK
    hs, ws = size[0] // 16, size[1] // 16
    names = ["a_tracker_in", "a_tracker1", "a_tracker2", "a_tracker3_1", "a_tracker3_2", "a_tracker4_1", "a_tracker4_2",
             "a_tracker5", "a_tracker6", "a_tracker7", "a_tracker_pred"]

    def exps(model):
        return [int(torch.floor(torch.log2(getattr(model, n).scale)).item()) for n in names]

    def scales(model):
        return [float(getattr(model, n).scale.item()) for n in names]
    for order in ("script", "prequant"):
        weights = synth.make_weights(**cfg["weights"], num_classes=classes)
        model = ref.Q("cpu", input_size=size, num_classes=classes, trainable=True, conf_thresh=0.01, nms_thresh=0.5,
                      anchor_size=anchors)
        G.load_weights(model, weights)
        model.train()
        q = ref.rbq
        q.quantized_layers.clear()
        if order == "prequant":
            q.init_quantize_net(model, 8)
            q.quantize_layers(8)
        sa_hist, sc_hist = [], []
        for it, seed in enumerate(cfg["seeds"]):
            x = synth.make_images(seed, cfg["batch"], size[0], size[1])
            target = torch.zeros(cfg["batch"], hs * ws * len(anchors), 11)
            try:
                with torch.no_grad():
                    model(torch.from_numpy(x), target=target, quantization=True)        # retune_bias_quantize.py:358
            except Exception as e:       # the loss of the training branch is irrelevant here; the trackers ran before it
                print("  (training-branch tail raised %s: ignored)" % type(e).__name__)
            q.init_quantize_net(model, 8)                                             # :361
            q.quantize_layers(8)                                                      # :362
            sa_hist.append(exps(model))
            sc_hist.append(scales(model))
        out["ema/%s/sa" % order] = np.array(sa_hist, np.int32)
        out["ema/%s/scale" % order] = np.array(sc_hist, np.float32)
        print("ema", order, sa_hist[0], "->", sa_hist[-1])
    out["ema/meta"] = np.array(size + [classes, cfg["batch"]] + cfg["seeds"], np.int64)


EVAL = dict(weights=dict(seed=2, pred_gain=400.0, obj_bias=-4.0), size=[240, 320], classes=20, seeds=[51, 52, 53, 54, 59],
            sizes=[(640, 480), (500, 375), (333, 500), (1280, 720), (320, 240)], conf=0.1)


def gen_evaluator(ref, out):
    """The reference's own evaluator loop (utils/vocapi_evaluator_mask.py:49-82) on a stub dataset: a fresh (un-calibrated)
    q_bf model, quantization=True; the first image calibrates the trackers, every image is forwarded alone, boxes are
    rescaled on the host.  Stored: all_boxes[cls][image].  The mAP code behind it needs the dataset files and is skipped."""
    import importlib
    import tempfile
    ev_mod = importlib.import_module("utils.vocapi_evaluator_mask")
    cfg = EVAL
    size, classes = cfg["size"], cfg["classes"]
    x = np.concatenate([synth.make_images(s, 1, size[0], size[1], "blocks") for s in cfg["seeds"]])

    class Stub:
        def __len__(self):
            return len(x)

        def pull_item(self, i):
            w, h = cfg["sizes"][i]
            return torch.from_numpy(x[i]), None, h, w
    model = G.build_q_model(ref, synth.make_weights(**cfg["weights"], num_classes=classes), classes, ref.config.ANCHOR_SIZE,
                            size, cfg["conf"], ref.rbq)
    ev = object.__new__(ev_mod.VOCAPIEvaluator_mask)          # __init__ opens the dataset files
    ev.dataset, ev.labelmap, ev.device = Stub(), list(range(classes)), "cpu"
    ev.output_dir = tempfile.mkdtemp()
    ev.map = 0.0
    ev.evaluate_detections = lambda boxes: None
    with torch.no_grad():
        ev.evaluate(model, quantization=True, find=False)
    tot = 0
    for j in range(classes):
        for i in range(len(x)):
            a = np.asarray(ev.all_boxes[j][i], np.float32).reshape(-1, 5)
            out["eval/boxes/%d/%d" % (j, i)] = a
            tot += len(a)
    out["eval/meta"] = np.array(size + [classes] + cfg["seeds"], np.int64)
    print("eval: %d detections over %d images" % (tot, len(x)))


def main():
    ref = G.import_reference()
    out = {}
    gen_evaluator(ref, out)
    gen_qf32(ref, out)
    gen_relu(ref, out)
    gen_base_transform(ref, out)
    gen_ema(ref, out)
    path = os.path.join(HERE, "r2.npz")
    np.savez_compressed(path, **out)
    print("r2.npz", os.path.getsize(path))


if __name__ == "__main__":
    main()
