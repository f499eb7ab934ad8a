"""End-to-end fixture table shared by the oracle and the GPU parity tests.
tag -> (synth.make_weights kwargs, anchors, image pattern); sizes/seeds live in the .npz."""
from yolo355 import synth

E2E = {
    "c1": (dict(seed=2), synth.ANCHOR_SIZE_MASK, "noise"),
    "sparse": (dict(seed=2, pred_gain=400.0, obj_bias=-4.0), synth.ANCHOR_SIZE_MASK, "noise"),
    "diverse": (dict(seed=2, pred_gain=400.0, obj_bias=-4.0), synth.ANCHOR_SIZE, "blocks"),
    "find": (dict(seed=2), synth.ANCHOR_SIZE_MASK, "noise"),
    "gap": (dict(seed=3, bias_gain=40.0, weight_gain=3.0), synth.ANCHOR_SIZE, "noise"),
    "batch": (dict(seed=2), synth.ANCHOR_SIZE_MASK, "noise"),
}

# fp32 model families: (tag, arch, [H, W], classes, weight seed, image seeds, pattern,
# pred_gain, obj_bias); goldens in tests/golden/fp32.npz (gen_golden_fp32.py)
FP32_CASES = [
    ("slim416", "slim_yolo_v2", [416, 416], 2, 5, [0], "noise", 1.5, -2.0),
    ("slim_voc", "slim_yolo_v2", [320, 416], 20, 6, [3], "blocks", 1.5, -2.0),
    ("slim_b2", "slim_yolo_v2", [96, 160], 2, 5, [7, 8], "noise", 1.5, -2.0),
    ("tiny416", "tiny_yolo_v3", [416, 416], 20, 7, [0], "blocks", 1.5, -2.0),
    ("tiny_b2", "tiny_yolo_v3", [224, 320], 3, 8, [4, 5], "noise", 1.5, -2.0),
]


def fp32_anchors(arch, classes):
    if arch == "tiny_yolo_v3":
        return synth.TINY_MULTI_ANCHOR_SIZE
    return synth.ANCHOR_SIZE_MASK if classes == 2 else synth.ANCHOR_SIZE


def fp32_setup(case):
    """(layers, anchors, A, x) of a FP32_CASES row, regenerated from the seeds."""
    import numpy as np
    tag, arch, size, classes, wseed, iseeds, pattern, pg, ob = case
    anchors = fp32_anchors(arch, classes)
    A = len(anchors) if arch == "slim_yolo_v2" else len(anchors) // 2
    layers = synth.make_fp32_model(arch, wseed, classes, A, pred_gain=pg, obj_bias=ob)
    x = np.concatenate([synth.make_images(s, 1, size[0], size[1], pattern=pattern) for s in iseeds])
    return layers, anchors, A, x
