#!/usr/bin/env python3
"""Golden vectors of the fp32 model families, produced by running the REFERENCE's
own SlimYOLOv2 (models/slim_yolo_v2.py:385-622, BN un-fused, eval) and YOLOv3tiny
(models/tiny_yolo_v3.py) classes in the build container.  Inputs / parameters come from the
build-owned generator (yolo355.synth), so only seeds and the reference's outputs are stored.

    python tests/golden/gen_golden_fp32.py        # rewrites tests/golden/fp32.npz
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (import recipe of the reference)

import torch  # noqa: E402
from yolo355 import synth  # noqa: E402

sys.path.insert(0, os.path.dirname(HERE))
from cases import FP32_CASES as CASES  # noqa: E402


def anchors_of(arch, classes, ref_cfg):
    if arch == "tiny_yolo_v3":
        return ref_cfg.TINY_MULTI_ANCHOR_SIZE
    return ref_cfg.ANCHOR_SIZE_MASK if classes == 2 else ref_cfg.ANCHOR_SIZE


def main():
    ref = G.import_reference()
    tiny_mod = importlib.import_module("models.tiny_yolo_v3")
    assert ref.config.TINY_MULTI_ANCHOR_SIZE == synth.TINY_MULTI_ANCHOR_SIZE
    out = {}
    for tag, arch, size, classes, wseed, iseeds, pattern, pg, ob in CASES:
        anchors = anchors_of(arch, classes, ref.config)
        A = len(anchors) if arch == "slim_yolo_v2" else len(anchors) // 2
        layers = synth.make_fp32_model(arch, wseed, classes, A, pred_gain=pg, obj_bias=ob)
        cls = ref.SlimYOLOv2 if arch == "slim_yolo_v2" else tiny_mod.YOLOv3tiny
        model = cls("cpu", input_size=size, num_classes=classes, trainable=False, conf_thresh=0.01,
                    nms_thresh=0.5, anchor_size=anchors)
        sd = model.state_dict()
        for k, v in synth.state_dict_fp32(layers).items():
            assert k in sd and tuple(sd[k].shape) == v.shape, k
            sd[k] = torch.from_numpy(v.copy())
        model.load_state_dict(sd)
        model.eval()
        # record the prediction maps the head consumes
        grabbed = {}
        hooks = []
        for name in (["pred"] if arch == "slim_yolo_v2" else ["pred_1", "pred_2"]):
            hooks.append(getattr(model, name).register_forward_hook(
                lambda m, i, o, n=name: grabbed.__setitem__(n, o.detach().numpy().copy())))
        x = np.concatenate([synth.make_images(s, 1, size[0], size[1], pattern=pattern) for s in iseeds])
        preds = {}
        for bi in range(x.shape[0]):
            for conf in (0.01, 0.1):
                model.conf_thresh = conf
                with torch.no_grad():
                    b, s, c = model(torch.from_numpy(x[bi:bi + 1]))
                out["%s/%d/det%g/boxes" % (tag, bi, conf)] = b.astype(np.float32)
                out["%s/%d/det%g/scores" % (tag, bi, conf)] = s.astype(np.float32)
                out["%s/%d/det%g/cls" % (tag, bi, conf)] = c.astype(np.int64)
            for n, v in grabbed.items():
                preds.setdefault(n, []).append(v)
        for n, v in preds.items():
            out["%s/%s" % (tag, n)] = np.concatenate(v).astype(np.float32)
        out["%s/meta" % tag] = np.array([size[0], size[1], classes, wseed] + list(iseeds), np.int64)
        for h in hooks:
            h.remove()
        print(tag, {n: v[0].shape for n, v in preds.items()}, "dets", [len(out["%s/%d/det0.01/scores" % (tag, i)]) for i in range(x.shape[0])],
              [len(out["%s/%d/det0.1/scores" % (tag, i)]) for i in range(x.shape[0])])
    path = os.path.join(HERE, "fp32.npz")
    np.savez_compressed(path, **out)
    print("fp32.npz", os.path.getsize(path))


if __name__ == "__main__":
    main()
