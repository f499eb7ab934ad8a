"""TEST INFRASTRUCTURE -- CPU restatement (numpy) of the reference's operator modules of the wider model
families.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
Pinned against tests/golden/ops.npz, which the reference's own modules produced (tests/golden/gen_golden_ops.py).

  reorg         utils.modules.reorg_layer.forward          utils/modules.py:48-57
  spp           utils.modules.SPP.forward                  utils/modules.py:66-72
  conv_bn_act   utils.modules.Conv2d.forward               utils/modules.py:6-18   (BN eval + LeakyReLU(0.125) / ReLU)
                backbone.darknet.Conv_BN_LeakyReLU.forward backbone/darknet.py:12-22 (LeakyReLU(0.1))
  resblock      backbone.darknet.resblock.forward          backbone/darknet.py:24-38
"""
import numpy as np


def reorg(x, stride):
    # utils/modules.py:50-55: view(B,C,h,s,w,s).transpose(3,4) -> view(B,C,h*w,s*s).transpose(2,3)
    # -> view(B,C,s*s,h,w).transpose(1,2) -> view(B,-1,h,w): out channel = (sy*s+sx)*C + c
    B, C, H, W = x.shape
    s = stride
    h, w = H // s, W // s
    y = x.reshape(B, C, h, s, w, s).transpose(0, 3, 5, 1, 2, 4)      # B, sy, sx, C, h, w
    return np.ascontiguousarray(y.reshape(B, s * s * C, h, w))


def _maxpool_same(x, k):
    # torch.nn.functional.max_pool2d(x, k, stride=1, padding=k//2): -inf padding (utils/modules.py:67-69)
    B, C, H, W = x.shape
    r = k // 2
    p = np.full((B, C, H + 2 * r, W + 2 * r), -np.inf, x.dtype)
    p[:, :, r:r + H, r:r + W] = x
    out = np.full_like(x, -np.inf)
    for dy in range(k):
        for dx in range(k):
            out = np.maximum(out, p[:, :, dy:dy + H, dx:dx + W])
    return out


def spp(x):
    return np.concatenate([x, _maxpool_same(x, 5), _maxpool_same(x, 9), _maxpool_same(x, 13)], axis=1)


def conv2d(x, w, b, stride=1):
    """nn.Conv2d with padding k//2 in float64."""
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    r = k // 2
    Ho = (H + 2 * r - k) // stride + 1
    Wo = (W + 2 * r - k) // stride + 1
    p = np.zeros((B, Cin, H + 2 * r, W + 2 * r), np.float64)
    p[:, :, r:r + H, r:r + W] = x
    out = np.zeros((B, Cout, Ho, Wo), np.float64)
    w64 = w.astype(np.float64)
    for dy in range(k):
        for dx in range(k):
            patch = p[:, :, dy:dy + stride * (Ho - 1) + 1:stride, dx:dx + stride * (Wo - 1) + 1:stride]
            out += np.einsum("bchw,oc->bohw", patch, w64[:, :, dy, dx])
    return out + b.astype(np.float64).reshape(1, -1, 1, 1)


def conv_bn_act(x, w, b, bn_w, bn_b, bn_mean, bn_var, stride=1, neg_slope=0.125, eps=1e-5):
    y = conv2d(x, w, b, stride)
    g = bn_w.astype(np.float64) / np.sqrt(bn_var.astype(np.float64) + eps)
    y = (y - bn_mean.astype(np.float64).reshape(1, -1, 1, 1)) * g.reshape(1, -1, 1, 1) + bn_b.astype(np.float64).reshape(1, -1, 1, 1)
    return np.where(y >= 0, y, y * neg_slope)


def resblock(x, blocks):
    """blocks: list of ((w0,b0,bn0), (w1,b1,bn1)), bn = (weight, bias, mean, var); slope 0.1 (darknet.py:18)."""
    x = x.astype(np.float64)
    for (w0, b0, bn0), (w1, b1, bn1) in blocks:
        y = conv_bn_act(x, w0, b0, *bn0, stride=1, neg_slope=0.1)
        y = conv_bn_act(y, w1, b1, *bn1, stride=1, neg_slope=0.1)
        x = y + x
    return x
