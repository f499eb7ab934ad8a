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
