"""TEST INFRASTRUCTURE -- fp32 CPU restatement (plain torch ops) of the reference's fp32 models:

    SlimYOLOv2.forward   models/slim_yolo_v2.py:549-601  (utils.modules.Conv2d :6-18 = conv+BN+LeakyReLU(0.125))
    YOLOv3tiny.forward   models/tiny_yolo_v3.py:176-243  (backbone DarkNet_Light, backbone/darknet.py:211-255)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this; the product
(yolo355/) never does.  Parity PINNED: tests/test_oracle_golden.py checks it against
tests/golden/fp32.npz, which tests/golden/gen_golden_fp32.py produced by running the reference's
own classes in the build container.  The HIP path computes in bf16 (fp32 accumulate); tests compare
it with this oracle within the tolerances stated in tests/test_gpu_fp32_models.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import yolo_oracle as O

EPS = 1e-5      # nn.BatchNorm2d default


def _block(x, L, slope):
    """conv -> BatchNorm(eval) -> LeakyReLU(slope); bare conv when the layer has no BN."""
    w, b = torch.from_numpy(L["w"]), torch.from_numpy(L["b"])
    y = F.conv2d(x, w, b, stride=1, padding=w.shape[2] // 2)
    if L["bn"] is None:
        return y
    g, be, mu, var = (torch.from_numpy(a) for a in L["bn"])
    y = F.batch_norm(y, mu, var, g, be, False, 0.1, EPS)
    return F.leaky_relu(y, slope)


def slim_preds(layers, x):
    """models/slim_yolo_v2.py:551-567.  Returns (pred [B,A(5+C),H/16,W/16], taps)."""
    t = torch.as_tensor(x, dtype=torch.float32)
    taps = []
    pools = {0, 1, 3, 5}
    with torch.no_grad():
        for i in range(9):
            t = _block(t, layers[i], 0.125)
            if i in pools:
                t = F.max_pool2d(t, 2, 2)
            taps.append(t)
        pred = _block(t, layers[9], None)
    return pred, taps


def tiny_preds(layers, x):
    """backbone/darknet.py:238-253 + models/tiny_yolo_v3.py:176-200.  Returns ([pred_1, pred_2], taps)
    with taps in the tensor order of csrc/net.hip."""
    t = torch.as_tensor(x, dtype=torch.float32)
    with torch.no_grad():
        taps = []
        for i in range(4):
            t = F.max_pool2d(_block(t, layers[i], 0.1), 2, 2)
            taps.append(t)
        c4 = _block(t, layers[4], 0.1)
        t5 = F.max_pool2d(c4, 2, 2)
        t6 = _block(t5, layers[5], 0.1)
        t7 = F.max_pool2d(F.pad(t6, (0, 1, 0, 1)), 2, 1)
        c5 = _block(t7, layers[6], 0.1)
        t9 = _block(c5, layers[7], 0.125)
        t10 = _block(t9, layers[8], 0.125)
        up = F.interpolate(t10, scale_factor=2.0, mode="bilinear", align_corners=True)
        cat = torch.cat([c4, up], dim=1)
        t11 = _block(cat, layers[9], 0.125)
        t12 = _block(t9, layers[10], 0.125)
        pred_2 = _block(t12, layers[11], None)
        pred_1 = _block(t11, layers[12], None)
        taps += [cat, t5, t6, t7, c5, t9, t10, t11, t12]
    return [pred_1, pred_2], taps


def tiny_head_decode(preds, input_size, anchors, num_classes, level_strides=(16, 32)):
    """models/tiny_yolo_v3.py:41-112, 202-232 (level_strides (16, 32)) and models/yolo_v3.py:65-110, 217-270
    (level_strides (8, 16, 32)) for every image: (bbox [B,N,4], cls_scores [B,N,C]).
    anchors: nlev*A pairs in pixels, finest level first."""
    nlev = len(level_strides)
    A = len(anchors) // nlev
    C = num_classes
    anc = torch.tensor(anchors, dtype=torch.float32).view(nlev, A, 2)
    w, h = input_size[1], input_size[0]
    confs, clss, txs, grids, strides, awh = [], [], [], [], [], []
    for ind, (pred, s) in enumerate(zip(preds, level_strides)):
        pred = torch.as_tensor(pred, dtype=torch.float32)
        B, abC, H, W = pred.shape
        p = pred.permute(0, 2, 3, 1).contiguous().view(B, H * W, abC)
        confs.append(p[:, :, :A].contiguous().view(B, H * W * A, 1))
        clss.append(p[:, :, A:(1 + C) * A].contiguous().view(B, H * W * A, C))
        txs.append(p[:, :, (1 + C) * A:].contiguous())
        ws, hs = w // s, h // s
        gy, gx = torch.meshgrid([torch.arange(hs), torch.arange(ws)], indexing="ij")
        grids.append(torch.stack([gx, gy], dim=-1).float().view(1, hs * ws, 1, 2))
        strides.append(torch.ones([1, hs * ws, A, 2]) * s)
        awh.append(anc[ind].repeat(hs * ws, 1, 1))
    conf = torch.cat(confs, 1)
    cls = torch.cat(clss, 1)
    B = conf.shape[0]
    HW = sum(g.shape[1] for g in grids)
    tx = torch.cat(txs, 1).view(B, HW, A, 4)
    grid, stride, anchor_wh = torch.cat(grids, 1), torch.cat(strides, 1), torch.cat(awh, 0).unsqueeze(0)
    cxy = (torch.sigmoid(tx[..., :2]) + grid) * stride
    bwh = torch.exp(tx[..., 2:]) * anchor_wh
    xywh = torch.cat([cxy, bwh], -1).view(B, HW * A, 4)
    box = torch.zeros_like(xywh)
    box[:, :, 0] = xywh[:, :, 0] - xywh[:, :, 2] / 2
    box[:, :, 1] = xywh[:, :, 1] - xywh[:, :, 3] / 2
    box[:, :, 2] = xywh[:, :, 0] + xywh[:, :, 2] / 2
    box[:, :, 3] = xywh[:, :, 1] + xywh[:, :, 3] / 2
    scale = torch.tensor([[[w, h, w, h]]]).float()
    box = torch.clamp(box / scale, 0., 1.)
    obj = torch.sigmoid(conf)
    sc = torch.softmax(cls, dim=2) * obj
    return box.numpy(), sc.numpy()


def detect(arch, layers, x, input_size, anchors, num_classes, conf_thresh=0.01, nms_thresh=0.5):
    """Whole fp32 path for a batch.  Returns dict(preds, taps, box, cls_scores, dets)."""
    if arch == "slim_yolo_v2":
        pred, taps = slim_preds(layers, x)
        preds = [pred]
        box, sc = O.head_decode(pred.numpy(), input_size, anchors, num_classes)
    else:
        preds, taps = tiny_preds(layers, x)
        box, sc = tiny_head_decode(preds, input_size, anchors, num_classes)
    dets = [O.postprocess(box[i], sc[i], conf_thresh, nms_thresh, num_classes) for i in range(box.shape[0])]
    return dict(preds=[p.numpy() for p in preds], taps=[t.numpy() for t in taps], box=np.asarray(box),
                cls_scores=np.asarray(sc), dets=dets)


def head_decode_v2(pred_f32, input_size, anchors, num_classes, stride):
    """models/yolo_v2.py:40-93, 183-204 (create_grid with ws = w // stride, decode_xywh * stride, / scale, clamp,
    sigmoid(conf) * softmax(cls)) for every image.  pred: [B, A*(5+C), H, W].  Returns (bbox [B,N,4], scores [B,N,C])."""
    pred = torch.as_tensor(pred_f32, dtype=torch.float32)
    A, C = len(anchors), num_classes
    B, abC, H, W = pred.shape
    w, h = input_size[1], input_size[0]
    ws, hs = w // stride, h // stride
    assert (hs, ws) == (H, W)
    p = pred.permute(0, 2, 3, 1).contiguous().view(B, H * W, abC)
    conf = p[:, :, :A].contiguous().view(B, H * W * A, 1)
    cls = p[:, :, A:(1 + C) * A].contiguous().view(B, H * W * A, C)
    txty = p[:, :, (1 + C) * A:].contiguous().view(B, H * W, A, 4)
    gy, gx = torch.meshgrid([torch.arange(hs), torch.arange(ws)], indexing="ij")
    grid_xy = torch.stack([gx, gy], dim=-1).float().view(1, hs * ws, 1, 2)
    anchor_wh = torch.tensor(anchors, dtype=torch.float32).repeat(hs * ws, 1, 1).unsqueeze(0)
    xy = torch.sigmoid(txty[..., :2]) + grid_xy
    wh = torch.exp(txty[..., 2:]) * anchor_wh
    xywh = torch.cat([xy, wh], -1).view(B, H * W * A, 4) * stride
    box = torch.zeros_like(xywh)
    box[:, :, 0] = xywh[:, :, 0] - xywh[:, :, 2] / 2
    box[:, :, 1] = xywh[:, :, 1] - xywh[:, :, 3] / 2
    box[:, :, 2] = xywh[:, :, 0] + xywh[:, :, 2] / 2
    box[:, :, 3] = xywh[:, :, 1] + xywh[:, :, 3] / 2
    scale = torch.tensor([[[w, h, w, h]]]).float()
    box = torch.clamp(box / scale, 0., 1.)
    return box.numpy(), (torch.softmax(cls, 2) * torch.sigmoid(conf)).numpy()


def detect_v2(pred_f32, anchors, num_classes, input_size, stride, conf_thresh, nms_thresh):
    """decode + postprocess (models/yolo_v2.py:137-163 = slim_yolo_v2.py:176-210) of a prediction map, per image."""
    box, sc = head_decode_v2(pred_f32, input_size, anchors, num_classes, stride)
    return [O.postprocess(box[i], sc[i], conf_thresh, nms_thresh, num_classes) for i in range(box.shape[0])]


def detect_v3(preds, anchors, num_classes, input_size, conf_thresh, nms_thresh):
    """decode + postprocess of the three prediction maps of yolo_v3 / yolo_v3_spp (models/yolo_v3.py:217-270), per image."""
    box, sc = tiny_head_decode(preds, input_size, anchors, num_classes, level_strides=(8, 16, 32))
    return [O.postprocess(box[i], sc[i], conf_thresh, nms_thresh, num_classes) for i in range(box.shape[0])]
