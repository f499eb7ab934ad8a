"""CPU oracle for the quantized slim-YOLOv2 hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*: a CPU restatement (numpy + stock torch CPU ops) of what the
reference computes on the path `SlimYOLOv2_quantize_bnfuse.forward(x, quantization=True)`.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the
product package (yolo355) never does.

Parity status: PINNED.  tests/golden/gen_golden.py imports the reference itself (in the
build container only), runs it on inputs from the build-owned generator and commits the
outputs under tests/golden/*.npz; tests/test_oracle_golden.py checks every function here
against those vectors.  The reference ships no tests or golden vectors of its own
 and its C path cannot be compiled (section 8c).

Reference sites restated (paths relative to the reference repo):
  * activation fake-quant + tracker state ... models/slim_yolo_v2.py:9-38
  * weight / bias pow2 quantizers ........... retune_bias_quantize.py:73-119
  * retune exponents ........................ retune_bias_quantize_findbest.py:115-148,
                                              c_embedding/yolo_forward.c:35
  * BN folding .............................. utils/bn_fuse.py:21-45
  * backbone schedule ....................... models/slim_yolo_v2.py:212-328,
                                              c_embedding/yolo_forward.c:1202-1262
  * shift composition ....................... c_embedding/yolo_forward.c:233-257
  * head split / decode / score ............. models/slim_yolo_v2.py:91-143,330-354
  * postprocess + greedy NMS ................ models/slim_yolo_v2.py:145-210
"""
import numpy as np
import torch
import torch.nn.functional as F

RETUNE = [11, 10, 10, 11, 11, 10, 11, 11, 11, 10]   # retune_bias_quantize_findbest.py:122-141
POOL_AFTER = [True, True, False, True, False, True, False, False, False, False]
LEAKY = [True] * 9 + [False]                         # pred has no activation (:321)
STRIDE = 16                                          # models/slim_yolo_v2.py:52


# ----------------------------------------------------------------------------- exponents
def floor_log2_scale(max_abs_f32):
    """e = floor(log2(127 / max)) evaluated in fp32 with the same torch ops the reference
    uses (slim_yolo_v2.py:22-23,33 ; retune_bias_quantize.py:79,84)."""
    m = torch.as_tensor(max_abs_f32, dtype=torch.float32).reshape(())
    scale = (2 ** (8 - 1) - 1) / m
    return int(torch.floor(torch.log2(scale)).item()), float(scale.item())


def exponent_of_scale(scale_f32):
    s = torch.as_tensor(scale_f32, dtype=torch.float32).reshape(())
    return int(torch.floor(torch.log2(s)).item())


def round_pow2(x_f32, e):
    """round(2^e * x) with torch.round (half-to-even), fp32 in, int32 out."""
    t = torch.as_tensor(np.ascontiguousarray(x_f32), dtype=torch.float32)
    s = torch.tensor(2.0, dtype=torch.float32) ** torch.tensor(float(e), dtype=torch.float32)
    return torch.round(s * t).to(torch.int64).numpy().astype(np.int32)


def quantize_tensor_pow2(t_f32):
    """retune_bias_quantize.py:73-97 (per-tensor, channel_level=False): returns
    (q int32 array with |q|<=127, exponent e) such that the reference stores q / 2^e."""
    t = np.asarray(t_f32, dtype=np.float32)
    e, _ = floor_log2_scale(np.abs(t).max())
    return round_pow2(t, e), e


def quantize_layers(weights):
    """weights: list of (name, W fp32, b fp32) -> list of dicts (quantize_layers :111-119)."""
    out = []
    for name, w, b in weights:
        qw, ew = quantize_tensor_pow2(w)
        qb, eb = quantize_tensor_pow2(b)
        out.append(dict(name=name, q_w=qw, e_w=ew, q_b=qb, e_b=eb))
    return out


class RangeTracker:
    """State machine of AveragedRangeTracker (slim_yolo_v2.py:9-38) driven by max|a|."""

    def __init__(self, momentum=0.1):
        self.momentum = np.float32(momentum)
        self.scale = torch.zeros(1)
        self.first_a = 0

    def update(self, max_abs_f32, freeze):
        m = torch.as_tensor(max_abs_f32, dtype=torch.float32).reshape(())
        s = (2 ** (8 - 1) - 1) / m
        if self.first_a == 0:
            self.first_a = 1
            self.scale = self.scale + s
        elif freeze:
            pass
        else:
            self.scale = self.scale * (1 - 0.1) + s * 0.1
        return self.exponent()

    def exponent(self):
        return int(torch.floor(torch.log2(self.scale)).item())


# ----------------------------------------------------------------------------- BN folding
def fuse_conv_and_bn(w, b, gamma, beta, mean, var, eps=1e-5, corrected=False):
    """utils/bn_fuse.py:21-45.  Reference formula leaves conv.bias unscaled;
    corrected=True applies the mathematically right fold."""
    w = torch.as_tensor(w, dtype=torch.float32)
    cout = w.shape[0]
    g = torch.as_tensor(gamma, dtype=torch.float32)
    v = torch.as_tensor(var, dtype=torch.float32)
    mu = torch.as_tensor(mean, dtype=torch.float32)
    be = torch.as_tensor(beta, dtype=torch.float32)
    bc = torch.zeros(cout) if b is None else torch.as_tensor(b, dtype=torch.float32)
    w_bn = torch.diag(g.div(torch.sqrt(eps + v)))
    wf = torch.mm(w_bn, w.reshape(cout, -1)).reshape(w.shape)
    b_bn = be - g.mul(mu).div(torch.sqrt(v + eps))
    if corrected:
        bf = bc * g.div(torch.sqrt(v + eps)) + b_bn
    else:
        bf = bc + b_bn
    return wf.numpy(), bf.numpy()


# ----------------------------------------------------------------------------- integer conv
def rne_shift(t, sh):
    """t * 2^-sh rounded half-to-even, exact on int64."""
    t = t.astype(np.int64)
    if sh <= 0:
        return t << np.int64(-sh)
    sh = np.int64(sh)
    return (t + ((np.int64(1) << (sh - 1)) - 1) + ((t >> sh) & 1)) >> sh


def conv3x3_int(q_in, q_w):
    """exact integer 3x3/pad1 conv via float64 (sums < 2^53)."""
    x = torch.as_tensor(q_in.astype(np.float64))
    w = torch.as_tensor(q_w.astype(np.float64))
    y = F.conv2d(x, w, None, 1, 1)
    return y.numpy().astype(np.int64)


def conv_layer_int(q_in, q_w, q_b, sa_in, e_w, e_b, leaky):
    """Pre-requant fixed-point value t' and its exponent F' (value = t'/2^F').
    Restates leaky(conv(x_q, W_q) + b_q) of slim_yolo_v2.py:220 etc. exactly;
    the same composition the FPGA is programmed with (yolo_forward.c:233-257)."""
    acc = conv3x3_int(q_in, q_w)
    Fx = max(sa_in + e_w, e_b)
    t = (acc << np.int64(Fx - sa_in - e_w)) + (q_b.astype(np.int64) << np.int64(Fx - e_b))[None, :, None, None]
    if isinstance(leaky, str):
        assert leaky == "relu"
        t = np.maximum(t, 0)               # nn.ReLU of Conv2d_fuse(leakyReLU=False), utils/modules.py:26: exponent unchanged
    elif leaky:
        t = np.where(t >= 0, t * 8, t)     # LeakyReLU(0.125): scale by 8, exponent +3
        Fx += 3
    return t, Fx, acc


def maxpool2x2(q):
    b, c, h, w = q.shape
    return q.reshape(b, c, h // 2, 2, w // 2, 2).max(axis=(3, 5))


def forward_backbone_int(x_f32, qlayers, trackers, quant_freeze=True, find=False,
                         saturate=False, keep=True):
    """Integer restatement of slim_yolo_v2.py:212-328 with quantization=True.

    trackers: list of 11 RangeTracker (input, conv1..conv7, pred); updated in place with
    the reference's first-call / EMA / freeze semantics.
    Returns dict: sa (11 exponents used), maps (list of post-quant, post-pool int32 maps,
    unclamped unless saturate), pred_q, sat (per-tracker count of |q|>127), sat_out (per layer: the same count taken on the
    layer's OUTPUT map, i.e. after the 2x2 max of a pooled layer -- what a kernel that pools before it clamps can count), guard (per-layer
    max |t'| * 2^(r-F'), the quantity `find` compares with 2^15), acc_max.
    """
    x = np.asarray(x_f32, dtype=np.float32)
    sa = []
    sat = []
    e0 = trackers[0].update(np.abs(x).max(), quant_freeze)
    q = round_pow2(x, e0)
    sa.append(e0)
    sat.append(int((np.abs(q) > 127).sum()))
    if saturate:
        q = np.clip(q, -127, 127)
    maps, guard, acc_max, sat_out = [], [], [], []
    for k, L in enumerate(qlayers):
        t, Fx, acc = conv_layer_int(q, L["q_w"], L["q_b"], sa[k], L["e_w"], L["e_b"], LEAKY[k])
        acc_max.append(int(np.abs(acc).max()))
        tmax = int(np.abs(t).max())
        g = tmax * (2.0 ** (RETUNE[k] - Fx))
        guard.append(g)
        if find and g >= 2 ** 15:
            raise AssertionError("too high!!! layer %d: %g" % (k, g))   # slim_yolo_v2.py:223-226
        # max|y| as the fp32 value the reference sees (exact while tmax < 2^24)
        ymax = np.float32(tmax) * np.float32(2.0 ** (-Fx))
        ek = trackers[k + 1].update(ymax, quant_freeze)
        sa.append(ek)
        q = rne_shift(t, Fx - ek)
        sat.append(int((np.abs(q) > 127).sum()))
        sat_out.append(int((np.abs(maxpool2x2(q) if POOL_AFTER[k] else q) > 127).sum()))
        if saturate:
            q = np.clip(q, -127, 127)
        if POOL_AFTER[k]:
            q = maxpool2x2(q)
        q = q.astype(np.int32)
        if keep:
            maps.append(q)
    return dict(sa=sa, maps=maps, pred_q=q, sat=sat, sat_out=sat_out, guard=guard, acc_max=acc_max)


# ----------------------------------------------------------------------------- head
def create_grid(input_size, anchors):
    """slim_yolo_v2.py:91-103.  input_size = [H, W]."""
    w, h = input_size[1], input_size[0]
    ws, hs = round(w / STRIDE), round(h / STRIDE)
    gy, gx = torch.meshgrid([torch.arange(hs), torch.arange(ws)], indexing="ij")
    grid_xy = torch.stack([gx, gy], dim=-1).float().view(1, hs * ws, 1, 2)
    anchor_wh = torch.tensor(anchors).repeat(hs * ws, 1, 1).unsqueeze(0)
    return grid_xy, anchor_wh


def head_decode(pred_f32, input_size, anchors, num_classes):
    """slim_yolo_v2.py:330-350 for every image of the batch (the reference does [0] only).
    pred_f32: [B, A*(5+C), H, W] float32.  Returns (bbox [B,N,4], cls_scores [B,N,C])."""
    pred = torch.as_tensor(pred_f32, dtype=torch.float32)
    A = len(anchors)
    C = num_classes
    B, abC, H, W = pred.shape
    p = pred.permute(0, 2, 3, 1).contiguous().view(B, H * W, abC)
    conf = p[:, :, :A].contiguous().view(B, H * W * A, 1)
    cls = p[:, :, A:(1 + C) * A].contiguous().view(B, H * W * A, C)
    txty = p[:, :, (1 + C) * A:].contiguous().view(B, H * W, A, 4)
    grid_xy, anchor_wh = create_grid(input_size, anchors)
    scale = torch.tensor([[[input_size[1], input_size[0], input_size[1], input_size[0]]]]).float()
    xy = torch.sigmoid(txty[..., :2]) + grid_xy
    wh = torch.exp(txty[..., 2:]) * anchor_wh
    xywh = torch.cat([xy, wh], -1).view(B, H * W * A, 4) * STRIDE
    box = torch.zeros_like(xywh)
    box[:, :, 0] = xywh[:, :, 0] - xywh[:, :, 2] / 2
    box[:, :, 1] = xywh[:, :, 1] - xywh[:, :, 3] / 2
    box[:, :, 2] = xywh[:, :, 0] + xywh[:, :, 2] / 2
    box[:, :, 3] = xywh[:, :, 1] + xywh[:, :, 3] / 2
    box = torch.clamp(box / scale, 0., 1.)
    obj = torch.sigmoid(conf)
    scores = torch.softmax(cls, 2) * obj
    return box.numpy(), scores.numpy()


def nms(dets, scores, nms_thresh):
    """slim_yolo_v2.py:145-174 with the build-defined tie order (score desc, index asc);
    the reference's argsort()[::-1] is an unstable sort, so its tie order is undefined."""
    x1, y1, x2, y2 = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-scores, kind="stable")
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        xx1 = np.maximum(x1[i], x1[order[1:]])
        yy1 = np.maximum(y1[i], y1[order[1:]])
        xx2 = np.minimum(x2[i], x2[order[1:]])
        yy2 = np.minimum(y2[i], y2[order[1:]])
        w = np.maximum(np.float32(1e-28), xx2 - xx1)
        h = np.maximum(np.float32(1e-28), yy2 - yy1)
        inter = w * h
        with np.errstate(invalid="ignore", divide="ignore"):
            ovr = inter / (areas[i] + areas[order[1:]] - inter)
        inds = np.where(ovr <= np.float32(nms_thresh))[0]
        order = order[inds + 1]
    return keep


def postprocess(bbox, prob, conf_thresh, nms_thresh, num_classes):
    """slim_yolo_v2.py:176-210 for one image.  Output in anchor-index order."""
    cls_inds = np.argmax(prob, axis=1)
    scores = prob[(np.arange(prob.shape[0]), cls_inds)].copy()
    keep = np.where(scores >= np.float32(conf_thresh))
    anchor_idx = keep[0]
    bbox = bbox[keep]
    scores = scores[keep]
    cls_inds = cls_inds[keep]
    flag = np.zeros(len(bbox), dtype=np.int64)
    for c in range(num_classes):
        inds = np.where(cls_inds == c)[0]
        if len(inds) == 0:
            continue
        c_keep = nms(bbox[inds], scores[inds], nms_thresh)
        flag[inds[c_keep]] = 1
    k = np.where(flag > 0)
    return bbox[k], scores[k], cls_inds[k], anchor_idx[k]


def detect(x_f32, qlayers, trackers, input_size, anchors, num_classes, conf_thresh=0.01,
           nms_thresh=0.5, find=False, saturate=False, keep=False):
    """Whole path for a batch: list of (bboxes, scores, cls_inds, anchor_idx) per image."""
    r = forward_backbone_int(x_f32, qlayers, trackers, True, find, saturate, keep)
    pred_f = (r["pred_q"].astype(np.float32) * np.float32(2.0 ** (-r["sa"][10])))
    box, sc = head_decode(pred_f, input_size, anchors, num_classes)
    dets = [postprocess(box[i], sc[i], conf_thresh, nms_thresh, num_classes)
            for i in range(box.shape[0])]
    r.update(dets=dets, box=box, cls_scores=sc)
    return r
