"""Build-owned deterministic data generator (counter-based splitmix64).

Everything the benchmarks, the golden-vector script and the GPU tests feed to the
engine comes from here, so the GPU box regenerates inputs and weights bit-for-bit
without any file from the reference.

Shapes follow the reference model table, models/slim_yolo_v2.py:59-87.
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

# (name, cin, cout, pool_after, leaky) -- models/slim_yolo_v2.py:59-87
SLIM_LAYERS = [
    ("conv1", 3, 16, True, True),
    ("conv2", 16, 32, True, True),
    ("conv3_1", 32, 64, False, True),
    ("conv3_2", 64, 64, True, True),
    ("conv4_1", 64, 128, False, True),
    ("conv4_2", 128, 128, True, True),
    ("conv5", 128, 256, False, True),
    ("conv6", 256, 256, False, True),
    ("conv7", 256, 256, False, True),
    ("pred", 256, None, False, False),
]

# data/config.py:10-14 (grid units, w,h)
ANCHOR_SIZE = [[1.19, 1.98], [2.79, 4.59], [4.53, 8.92], [8.06, 5.29], [10.32, 10.65]]
ANCHOR_SIZE_MASK = [[0.27894, 0.49337], [0.8669, 1.37835], [1.82727, 2.8404],
                    [3.4131, 5.05744], [5.8903, 7.6757]]
ANCHOR_SIZE_COCO = [[0.53, 0.79], [1.71, 2.36], [2.89, 6.44], [6.33, 3.79], [9.03, 9.74]]

# data/__init__.py:50 gives mean/std in BGR; test.py:79 swaps to RGB before the net
MEAN_RGB = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD_RGB = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def splitmix64(seed, n, offset=0):
    """n 64-bit words of the splitmix64 stream `seed`, starting at counter `offset`."""
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + i * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform_pm1(seed, shape):
    """float32 uniform in [-1, 1) with 24 random bits (exactly representable)."""
    n = int(np.prod(shape))
    z = splitmix64(seed, n)
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (u * 2.0 - 1.0).astype(np.float32).reshape(shape)


def uniform_u8(seed, shape):
    n = int(np.prod(shape))
    z = splitmix64(seed, n)
    return (z >> np.uint64(56)).astype(np.uint8).reshape(shape)


def make_frames_u8(seed, batch, h, w, pattern="noise"):
    """Synthetic camera frames, uint8 HWC BGR like cv2 would hand to BaseTransform.
    pattern "noise": iid uniform pixels.  "blocks": random 32x32 colour blocks plus
    +-16 noise, so that deep features (and hence detection scores) vary over the image."""
    if pattern == "noise":
        return uniform_u8(seed, (batch, h, w, 3))
    bh, bw = (h + 31) // 32, (w + 31) // 32
    coarse = uniform_u8(seed * 31 + 7, (batch, bh, bw, 3)).astype(np.int32)
    up = np.repeat(np.repeat(coarse, 32, axis=1), 32, axis=2)[:, :h, :w, :]
    noise = (uniform_u8(seed, (batch, h, w, 3)).astype(np.int32) >> 3) - 16
    return np.clip(up + noise, 0, 255).astype(np.uint8)


def normalize_frames(frames_u8_bgr):
    """BaseTransform (data/__init__.py:30-56) without the resize, then the BGR->RGB
    swap and HWC->CHW permute of test.py:79-80.  float32 NCHW out."""
    x = frames_u8_bgr.astype(np.float32)
    x /= np.float32(255.0)
    x -= MEAN_RGB[::-1]
    x /= STD_RGB[::-1]
    x = x[..., ::-1]
    return np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2)))


def make_images(seed, batch, h=416, w=416, pattern="noise"):
    """float32 NCHW batch emulating normalised 8-bit images."""
    return normalize_frames(make_frames_u8(seed, batch, h, w, pattern))


def pred_channels(num_classes, num_anchors=5):
    return num_anchors * (1 + 4 + num_classes)


# exponents of the trained FPGA model, recorded only in the C driver (c_embedding/yolo_forward.c:32-33)
C_TABLE_SCALE_W = [6, 8, 8, 9, 9, 9, 10, 10, 10, 9]
C_TABLE_SCALE_B = [7, 6, 5, 5, 5, 6, 5, 5, 5, 10]


def make_weights(seed, num_classes=2, num_anchors=5, bias_gain=1.0, weight_gain=1.0,
                 obj_bias=None, pred_gain=1.0, ctable=False):
    """fp32 (W[cout,cin,3,3], b[cout]) for the 10 convs, PyTorch-default-like
    U(+-1/sqrt(fan_in)).  `obj_bias` (float) overrides the objectness biases of the
    pred layer (the "sparse detections" fixture); `pred_gain` scales the
    pred weights (not biases) so that logits spread over the int8 range (few score ties).
    `ctable`: one weight and one bias per tensor are set to 0.9 * 127 / 2^e with e from the C driver's tables, so that
    the per-tensor power-of-two quantizer (retune_bias_quantize.py:73-119) lands on exactly those exponents -- like a
    trained, BN-folded tensor: a few large values, the bulk small."""
    out = []
    for li, (name, cin, cout, _pool, _act) in enumerate(SLIM_LAYERS):
        if cout is None:
            cout = pred_channels(num_classes, num_anchors)
        bound = 1.0 / np.sqrt(cin * 9.0)
        w = uniform_pm1(seed * 1000 + 2 * li, (cout, cin, 3, 3)) * np.float32(bound * weight_gain)
        b = uniform_pm1(seed * 1000 + 2 * li + 1, (cout,)) * np.float32(bound * bias_gain)
        if name == "pred":
            w = w * np.float32(pred_gain)
            if obj_bias is not None:
                b[:num_anchors] = np.float32(obj_bias)
        if ctable:
            w = w.copy()
            b = b.copy()
            tw, tb = 0.9 * 127.0 / 2.0 ** C_TABLE_SCALE_W[li], 0.9 * 127.0 / 2.0 ** C_TABLE_SCALE_B[li]
            w *= np.float32(min(1.0, 0.5 * tw / float(np.abs(w).max())))       # the bulk stays below the planted maximum
            b *= np.float32(min(1.0, 0.5 * tb / float(np.abs(b).max())))
            w.flat[(7 * li + 3) % w.size] = np.float32(tw if li % 2 == 0 else -tw)
            b.flat[(5 * li + 1) % b.size] = np.float32(-tb if li % 3 == 0 else tb)
        out.append((name, w.astype(np.float32), b.astype(np.float32)))
    return out


def make_bn(seed, cout):
    """BatchNorm2d eval statistics (gamma, beta, running_mean, running_var)."""
    g = 1.0 + 0.25 * uniform_pm1(seed * 7 + 1, (cout,))
    be = 0.1 * uniform_pm1(seed * 7 + 2, (cout,))
    mu = 0.1 * uniform_pm1(seed * 7 + 3, (cout,))
    var = 1.0 + 0.5 * uniform_pm1(seed * 7 + 4, (cout,))
    return (g.astype(np.float32), be.astype(np.float32), mu.astype(np.float32),
            var.astype(np.float32))


# ---- fp32 model families (SlimYOLOv2 with BatchNorm, YOLOv3tiny) ---------------------------
MULTI_ANCHOR_SIZE = [[32.64, 47.68], [50.24, 108.16], [126.72, 96.32], [78.4, 201.92], [178.24, 178.56],
                     [129.6, 294.72], [331.84, 194.56], [227.84, 325.76], [365.44, 358.72]]      # data/config.py:18-20 (yolo_v3, VOC)
TINY_MULTI_ANCHOR_SIZE = [[34.01, 61.79], [86.94, 109.68], [93.49, 227.46],
                          [246.38, 163.33], [178.68, 306.55], [344.89, 337.14]]      # data/config.py:27-28
# (state_dict prefix, cin, cout, ksize, has_bn) in the weight-slot order of csrc/net.hip
TINY_LAYERS = [
    ("backbone.conv_1", 3, 16, 3, True), ("backbone.conv_2", 16, 32, 3, True), ("backbone.conv_3", 32, 64, 3, True),
    ("backbone.conv_4", 64, 128, 3, True), ("backbone.conv_5", 128, 256, 3, True), ("backbone.conv_6", 256, 512, 3, True),
    ("backbone.conv_7", 512, 1024, 3, True), ("conv_set_2", 1024, 256, 3, True), ("conv_1x1_2", 256, 128, 1, True),
    ("conv_set_1", 384, 256, 3, True), ("extra_conv_2", 256, 512, 3, True), ("pred_2", 512, None, 1, False),
    ("pred_1", 256, None, 1, False),
]


def make_fp32_model(arch, seed, num_classes, num_anchors, weight_gain=2.2, pred_gain=1.0, obj_bias=None):
    """Synthetic fp32 parameters of an un-fused model: list of dicts
    {name, w [cout,cin,k,k], b [cout], bn (gamma, beta, mean, var) or None} in weight-slot order.
    weight_gain ~ 2.2 keeps activations O(1) through the LeakyReLU stack."""
    if arch == "slim_yolo_v2":
        table = [(n, cin, cout, 3, n != "pred") for (n, cin, cout, _p, _a) in SLIM_LAYERS]
    elif arch == "tiny_yolo_v3":
        table = TINY_LAYERS
    else:
        raise ValueError(arch)
    out = []
    for li, (name, cin, cout, k, has_bn) in enumerate(table):
        if cout is None:
            cout = pred_channels(num_classes, num_anchors)
        bound = 1.0 / np.sqrt(cin * k * k)
        gain = weight_gain if has_bn else pred_gain
        w = uniform_pm1(seed * 1000 + 2 * li, (cout, cin, k, k)) * np.float32(bound * gain)
        b = uniform_pm1(seed * 1000 + 2 * li + 1, (cout,)) * np.float32(bound)
        if not has_bn and obj_bias is not None:
            b[:num_anchors] = np.float32(obj_bias)
        out.append(dict(name=name, w=w.astype(np.float32), b=b.astype(np.float32),
                        bn=make_bn(seed * 100 + li, cout) if has_bn else None))
    return out


def state_dict_fp32(layers):
    """torch state_dict entries (numpy) for the reference's / the drop-ins' module names."""
    sd = {}
    for L in layers:
        n = L["name"]
        if L["bn"] is None:
            sd[n + ".weight"], sd[n + ".bias"] = L["w"], L["b"]
        else:
            g, be, mu, var = L["bn"]
            sd[n + ".convs.0.weight"], sd[n + ".convs.0.bias"] = L["w"], L["b"]
            sd[n + ".convs.1.weight"], sd[n + ".convs.1.bias"] = g, be
            sd[n + ".convs.1.running_mean"], sd[n + ".convs.1.running_var"] = mu, var
    return sd
