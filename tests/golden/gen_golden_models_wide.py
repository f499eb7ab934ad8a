#!/usr/bin/env python3
"""Golden vectors of the wider model families, produced by running the REFERENCE's own
myYOLOv2 (models/yolo_v2.py), myYOLOv3 (models/yolo_v3.py) and myYOLOv3Spp (models/yolo_v3_spp.py) in the build container, eval mode, CPU fp32.  Weights and inputs come from the
build-owned generator (tests/cases.py:synth_state_dict, yolo355.synth.make_images); stored: the prediction map
and the detections at conf 0.05.

    python tests/golden/gen_golden_models_wide.py        # rewrites tests/golden/models_wide.npz
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

import torch  # noqa: E402
from yolo355 import synth  # noqa: E402

sys.path.insert(0, os.path.dirname(HERE))
from cases import WIDE_MODEL_CASES, WIDE3_MODEL_CASES, synth_state_dict  # noqa: E402


def main():
    G.import_reference()
    out = {}
    for tag, cls, size, classes, seed in WIDE_MODEL_CASES:
        mod = importlib.import_module("models.yolo_v2")
        m = getattr(mod, cls)("cpu", input_size=size, num_classes=classes, trainable=False, conf_thresh=0.05, nms_thresh=0.5,
                              anchor_size=synth.ANCHOR_SIZE)
        m.load_state_dict(synth_state_dict(m.state_dict(), seed))
        m.eval()
        grabbed = {}
        m.pred.register_forward_hook(lambda mm, i, o: grabbed.__setitem__("pred", o.detach().numpy().copy()))
        x = torch.from_numpy(synth.make_images(seed + 1, 1, size[0], size[1]))
        with torch.no_grad():
            b, s, c = m(x)
        out[tag + "_pred"] = grabbed["pred"].astype(np.float32)
        out[tag + "_boxes"] = np.asarray(b, np.float32)
        out[tag + "_scores"] = np.asarray(s, np.float32)
        out[tag + "_cls"] = np.asarray(c, np.int64)
        print(tag, grabbed["pred"].shape, float(np.abs(grabbed["pred"]).max()), len(s))
    for tag, modname, cls, size, classes, seed, gain in WIDE3_MODEL_CASES:
        mod = importlib.import_module(modname)
        m = getattr(mod, cls)("cpu", input_size=size, num_classes=classes, trainable=False, conf_thresh=0.05, nms_thresh=0.5,
                              anchor_size=synth.MULTI_ANCHOR_SIZE)
        m.load_state_dict(synth_state_dict(m.state_dict(), seed, weight_gain=gain))
        m.eval()
        grabbed = {}
        for name in ("pred_1", "pred_2", "pred_3"):
            getattr(m, name).register_forward_hook(lambda mm, i, o, n=name: grabbed.__setitem__(n, o.detach().numpy().copy()))
        x = torch.from_numpy(synth.make_images(seed + 1, 1, size[0], size[1]))
        with torch.no_grad():
            b, s, c = m(x)
        for name in ("pred_1", "pred_2", "pred_3"):
            out[tag + "_" + name] = grabbed[name].astype(np.float16 if name == "pred_1" else np.float32)   # the 28x28 map is the big one
        out[tag + "_boxes"] = np.asarray(b, np.float32)
        out[tag + "_scores"] = np.asarray(s, np.float32)
        out[tag + "_cls"] = np.asarray(c, np.int64)
        print(tag, [grabbed[n].shape for n in grabbed], float(np.abs(grabbed["pred_3"]).max()), len(s))
    # state_dict layouts (key order and shapes) of the three classes: what "checkpoints load unchanged" means
    import json
    layouts = {}
    for modname, cls, anchors in (("models.yolo_v2", "myYOLOv2", synth.ANCHOR_SIZE), ("models.yolo_v3", "myYOLOv3", synth.MULTI_ANCHOR_SIZE),
                                  ("models.yolo_v3_spp", "myYOLOv3Spp", synth.MULTI_ANCHOR_SIZE)):
        m = getattr(importlib.import_module(modname), cls)("cpu", input_size=[224, 224], num_classes=20, trainable=False,
                                                           anchor_size=anchors)
        layouts[cls] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    with open(os.path.join(HERE, "models_wide_layout.json"), "w") as f:
        json.dump(layouts, f)
    np.savez_compressed(os.path.join(HERE, "models_wide.npz"), **out)


if __name__ == "__main__":
    main()
