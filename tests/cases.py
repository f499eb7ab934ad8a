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
    # round 3: weights that quantize to the trained model's exponents (c_embedding/yolo_forward.c:32-33), FPGA size 240 x 320
    "ctable": (dict(seed=4, ctable=True, pred_gain=3.0, obj_bias=-1.0), synth.ANCHOR_SIZE_MASK, "blocks"),
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


# ---- operator-level cases of the wider model families: (tag, kind, params)
#   reorg:  (B, C, H, W, stride)
#   spp:    (B, C, H, W)
#   conv:   (module, B, Cin, Cout, H, W, ksize, stride, leaky)   module in {"Conv2d", "Conv_BN_LeakyReLU"}
#   resblock: (B, ch, H, W, nblocks)
OPS_CASES = [
    ("reorg_2", "reorg", (2, 6, 8, 12, 2)),
    ("reorg_26", "reorg", (1, 64, 26, 26, 2)),
    ("spp_13", "spp", (2, 5, 13, 13)),
    ("spp_7x9", "spp", (1, 3, 7, 9)),
    ("conv_3x3_l", "conv", ("Conv2d", 2, 40, 70, 13, 13, 3, 1, True)),
    ("conv_1x1_relu", "conv", ("Conv2d", 1, 64, 32, 26, 26, 1, 1, False)),
    ("conv_thin", "conv", ("Conv_BN_LeakyReLU", 1, 3, 32, 32, 48, 3, 1, True)),
    ("conv_s2", "conv", ("Conv_BN_LeakyReLU", 2, 32, 64, 32, 32, 3, 2, True)),
    ("conv_s2_odd", "conv", ("Conv_BN_LeakyReLU", 1, 64, 96, 15, 21, 3, 2, True)),
    ("resblock_64", "resblock", (1, 64, 16, 16, 2)),
]


def ops_inputs(tag, kind, prm):
    """Deterministic operands of an operator case from the build-owned generator (no torch RNG)."""
    import numpy as np
    from yolo355 import synth
    seed = 7000 + sum(ord(ch) for ch in tag)
    if kind == "reorg":
        B, C, H, W, s = prm
        return {"x": synth.uniform_pm1(seed, (B, C, H, W)).astype(np.float32)}
    if kind == "spp":
        B, C, H, W = prm
        return {"x": synth.uniform_pm1(seed, (B, C, H, W)).astype(np.float32)}
    if kind == "conv":
        mod, B, Cin, Cout, H, W, k, s, leaky = prm
        fan = Cin * k * k
        d = {"x": synth.uniform_pm1(seed, (B, Cin, H, W)).astype(np.float32),
             "w": (synth.uniform_pm1(seed + 1, (Cout, Cin, k, k)) * (2.0 / np.sqrt(fan))).astype(np.float32),
             "b": (synth.uniform_pm1(seed + 2, (Cout,)) * 0.2).astype(np.float32)}
        g, be, mu, var = synth.make_bn(seed + 3, Cout)
        d.update(bn_w=g, bn_b=be, bn_mean=mu, bn_var=var)
        return d
    B, ch, H, W, nb = prm
    d = {"x": synth.uniform_pm1(seed, (B, ch, H, W)).astype(np.float32)}
    for i in range(nb):
        for j, (ci, co, k) in enumerate([(ch, ch // 2, 1), (ch // 2, ch, 3)]):
            s2 = seed + 10 * (2 * i + j + 1)
            d["w%d_%d" % (i, j)] = (synth.uniform_pm1(s2, (co, ci, k, k)) * (2.0 / np.sqrt(ci * k * k))).astype(np.float32)
            d["b%d_%d" % (i, j)] = (synth.uniform_pm1(s2 + 1, (co,)) * 0.2).astype(np.float32)
            g, be, mu, var = synth.make_bn(s2 + 2, co)
            d["bn%d_%d" % (i, j)] = np.stack([g, be, mu, var]).astype(np.float32)
    return d


def synth_state_dict(sd, seed, weight_gain=2.0):
    """Fill a state_dict (reference model or drop-in: same keys) from the build-owned generator, by key and shape:
    conv weights U(+-gain/sqrt(fan_in)), biases U(+-0.1), BatchNorm eval statistics from synth.make_bn."""
    import numpy as np
    import torch
    from yolo355 import synth
    out = {}
    bn_cache = {}
    for i, (k, v) in enumerate(sd.items()):
        shape = tuple(v.shape)
        s = seed + 13 * i
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
        elif k.endswith(".weight") and len(shape) == 4:
            fan = shape[1] * shape[2] * shape[3]
            gain = weight_gain * (8.0 if k.startswith("pred") else 1.0)      # lively logits: boxes and scores that differ
            out[k] = torch.from_numpy((synth.uniform_pm1(s, shape) * (gain / np.sqrt(fan))).astype(np.float32))
        elif len(shape) == 1 and (k.endswith("running_var") or k.endswith("running_mean") or ".1." in k):
            base = k.rsplit(".", 1)[0]
            if base not in bn_cache:
                bn_cache[base] = synth.make_bn(seed + 7 * len(bn_cache) + 1, shape[0])
            g, be, mu, var = bn_cache[base]
            out[k] = torch.from_numpy({"weight": g, "bias": be, "running_mean": mu, "running_var": var}[k.rsplit(".", 1)[1]].copy())
        else:
            out[k] = torch.from_numpy((synth.uniform_pm1(s, shape) * 0.1).astype(np.float32))
    return out


# whole-model cases of the wider families, composed from the operator API: (tag, class, input size, classes, seed)
WIDE_MODEL_CASES = [("yolo_v2_224", "myYOLOv2", [224, 224], 20, 4100)]
# three-level models: (tag, reference module, class, input size, classes, seed, weight gain); anchors = synth.MULTI_ANCHOR_SIZE
# (gain 1.3 keeps the activations of the 75 residual-connected layers O(1): logits within +-2)
WIDE3_MODEL_CASES = [("yolo_v3_224", "models.yolo_v3", "myYOLOv3", [224, 224], 20, 4200, 1.3),
                     ("yolo_v3_spp_224", "models.yolo_v3_spp", "myYOLOv3Spp", [224, 224], 20, 4300, 1.3)]
