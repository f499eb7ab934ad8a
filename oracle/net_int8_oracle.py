"""TEST INFRASTRUCTURE -- integer CPU restatement of the int8 form of the y355_net graphs
(csrc/net.hip): SlimYOLOv2 topology and YOLOv3tiny (models/tiny_yolo_v3.py:176-243,
backbone/darknet.py:238-253) under the q_bf recipe of the reference
(retune_bias_quantize.py:73-119 weights, models/slim_yolo_v2.py:16-38 activations).

PARITY UNPINNED for the int8 YOLOv3tiny: the reference has no int8 form of this model (SURVEY.md
8a-17), so nothing of the reference pins these integer semantics.  They are build-defined
(DESIGN.md "int8 YOLOv3tiny"), restated here independently of the HIP code, and additionally
held to the reference's fp32 outputs (tests/golden/fp32.npz) within a stated tolerance.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.

Semantics per conv (same as oracle/yolo_oracle.py, with a general LeakyReLU slope m / 2^lk):
    acc = sum q_a q_w;  F = max(sa_in + e_w, e_b);  t = acc << (F - sa_in - e_w) + q_b << (F - e_b)
    t' = t >= 0 ? t << lk : t * m      (0.125: lk 3, m 1;  0.1: lk 11, m 205;  none: lk 0, m 1)
    q  = clamp(RNE(t' * 2^(sa_out - F - lk)), +-127);  2x2 max-pool on q
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import fp32_oracle as FP
from . import yolo_oracle as O

ACT = {None: (0, 1), 0.125: (3, 1), 0.1: (11, 205)}

# (op, in, out, choff, layer, ksize, pool, slope) -- mirrors the graph of models/tiny_yolo_v3.py
TINY_OPS = [
    ("conv", -1, 0, 0, 0, 3, 1, 0.1), ("conv", 0, 1, 0, 1, 3, 1, 0.1), ("conv", 1, 2, 0, 2, 3, 1, 0.1),
    ("conv", 2, 3, 0, 3, 3, 1, 0.1), ("conv", 3, 4, 0, 4, 3, 0, 0.1), ("pool", 4, 5, 0, -1, 2, 2, None),
    ("conv", 5, 6, 0, 5, 3, 0, 0.1), ("pool", 6, 7, 0, -1, 2, 1, None), ("conv", 7, 8, 0, 6, 3, 0, 0.1),
    ("conv", 8, 9, 0, 7, 3, 0, 0.125), ("conv", 9, 10, 0, 8, 1, 0, 0.125), ("up", 10, 4, 256, -1, 0, 0, None),
    ("conv", 4, 11, 0, 9, 3, 0, 0.125), ("conv", 9, 12, 0, 10, 3, 0, 0.125), ("conv", 12, 14, 0, 11, 1, 0, None),
    ("conv", 11, 13, 0, 12, 1, 0, None),
]
TINY_CH = [16, 32, 64, 128, 384, 256, 512, 512, 1024, 256, 128, 256, 512, None, None]
TINY_POOL_IN_C = {5: 256, 7: 512}


def fold_bn(layers):
    """exact eval fold (float64, rounded to fp32) of the un-fused synthetic layers -> [(w, b)]"""
    out = []
    for L in layers:
        w, b = L["w"].astype(np.float64), L["b"].astype(np.float64)
        if L["bn"] is not None:
            g, be, mu, var = (a.astype(np.float64) for a in L["bn"])
            s = g / np.sqrt(var + FP.EPS)
            w, b = w * s[:, None, None, None], (b - mu) * s + be
        out.append((w.astype(np.float32), b.astype(np.float32)))
    return out


def quantize_folded(folded):
    """per-tensor power-of-two int8 of weights and biases (retune_bias_quantize.py:73-119)"""
    out = []
    for w, b in folded:
        qw, ew = O.quantize_tensor_pow2(w)
        qb, eb = O.quantize_tensor_pow2(b)
        out.append(dict(q_w=qw, e_w=ew, q_b=qb, e_b=eb))
    return out


def conv_int(q_in, q_w):
    x = torch.as_tensor(q_in.astype(np.float64))
    w = torch.as_tensor(q_w.astype(np.float64))
    return F.conv2d(x, w, None, 1, w.shape[2] // 2).numpy().astype(np.int64)


def conv_layer(q_in, sa_in, L, sa_out, slope, pool):
    acc = conv_int(q_in, L["q_w"])
    Fb = max(sa_in + L["e_w"], L["e_b"])
    t = acc * (np.int64(1) << np.int64(Fb - sa_in - L["e_w"])) + \
        (L["q_b"].astype(np.int64) * (np.int64(1) << np.int64(Fb - L["e_b"])))[None, :, None, None]
    lk, m = ACT[slope]
    t = np.where(t >= 0, t * (np.int64(1) << np.int64(lk)), t * np.int64(m))
    q = O.rne_shift(t, Fb + lk - sa_out)
    sat = int((np.abs(q) > 127).sum())
    q = np.clip(q, -127, 127)
    if pool:
        B, C, H, W = q.shape
        q = q.reshape(B, C, H // 2, 2, W // 2, 2).max(axis=(3, 5))
    return q, sat


def upsample_int(q_in, rescale):
    """the fp32 expression of csrc/net.hip upsample_i8_kernel, operation for operation"""
    B, C, H, W = q_in.shape
    Ho, Wo = 2 * H, 2 * W
    f = np.float32
    ry, rx = f(H - 1) / f(Ho - 1), f(W - 1) / f(Wo - 1)
    sy = ry * np.arange(Ho, dtype=np.float32)
    sx = rx * np.arange(Wo, dtype=np.float32)
    y0, x0 = sy.astype(np.int32), sx.astype(np.int32)
    y1, x1 = np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
    ly, lx = (sy - y0.astype(np.float32)).astype(np.float32), (sx - x0.astype(np.float32)).astype(np.float32)
    hy, hx = (f(1) - ly).astype(np.float32), (f(1) - lx).astype(np.float32)
    v = q_in.astype(np.float32)
    a, b_ = v[:, :, y0][:, :, :, x0], v[:, :, y0][:, :, :, x1]
    c, d = v[:, :, y1][:, :, :, x0], v[:, :, y1][:, :, :, x1]
    HX, LX = hx[None, None, None, :], lx[None, None, None, :]
    HY, LY = hy[None, None, :, None], ly[None, None, :, None]
    top = (HX * a).astype(np.float32) + (LX * b_).astype(np.float32)
    bot = (HX * c).astype(np.float32) + (LX * d).astype(np.float32)
    out = (HY * top.astype(np.float32)).astype(np.float32) + (LY * bot.astype(np.float32)).astype(np.float32)
    q = np.rint(out.astype(np.float32) * f(rescale))
    return np.clip(q, -127, 127).astype(np.int64)


def tiny_forward_int(x_f32, qlayers, sa_in, sa):
    """int8 YOLOv3tiny: returns dict(t = list of int tensors in graph order, sat = clamped count).
    sa: one exponent per tensor (pool outputs take their input's)."""
    sa = list(sa)
    x = np.asarray(x_f32, dtype=np.float32)
    r = np.rint(x * np.float32(2.0 ** sa_in))
    sat = int((np.abs(r) > 127).sum())
    q_x = np.clip(r, -127, 127).astype(np.int64)
    T = [None] * len(TINY_CH)
    for op, i, o, choff, li, k, pool, slope in TINY_OPS:
        if op == "conv":
            src = q_x if i < 0 else T[i]
            q, s = conv_layer(src, sa_in if i < 0 else sa[i], qlayers[li], sa[o], slope, pool == 1)
            sat += s
            if TINY_CH[o] is not None and q.shape[1] < TINY_CH[o]:        # first writer of the concat buffer
                buf = np.zeros((q.shape[0], TINY_CH[o]) + q.shape[2:], np.int64)
                buf[:, choff:choff + q.shape[1]] = q
                T[o] = buf
            else:
                T[o] = q
        elif op == "pool":
            src = T[i][:, :TINY_POOL_IN_C[o]]
            sa[o] = sa[i]
            if pool == 2:
                B, C, H, W = src.shape
                T[o] = src.reshape(B, C, H // 2, 2, W // 2, 2).max(axis=(3, 5))
            else:       # ZeroPad2d((0,1,0,1)) + MaxPool2d(2, 1)
                p = np.pad(src, ((0, 0), (0, 0), (0, 1), (0, 1)))
                T[o] = np.maximum(np.maximum(p[:, :, :-1, :-1], p[:, :, :-1, 1:]), np.maximum(p[:, :, 1:, :-1], p[:, :, 1:, 1:]))
        else:
            up = upsample_int(T[i], 2.0 ** (sa[o] - sa[i]))
            T[o][:, choff:choff + up.shape[1]] = up
    return dict(t=T, sat=sat, sa=sa)


def tiny_detect(x_f32, qlayers, sa_in, sa, input_size, anchors, num_classes, conf_thresh=0.01, nms_thresh=0.5):
    r = tiny_forward_int(x_f32, qlayers, sa_in, sa)
    preds = [r["t"][13].astype(np.float32) * np.float32(2.0 ** (-r["sa"][13])),
             r["t"][14].astype(np.float32) * np.float32(2.0 ** (-r["sa"][14]))]
    box, sc = FP.tiny_head_decode(preds, input_size, anchors, num_classes)
    r.update(box=np.asarray(box), cls_scores=np.asarray(sc), preds=preds,
             dets=[O.postprocess(box[i], sc[i], conf_thresh, nms_thresh, num_classes) for i in range(box.shape[0])])
    return r
