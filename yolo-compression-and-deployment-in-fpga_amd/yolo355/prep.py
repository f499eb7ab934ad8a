"""Host-side weight preparation of the quantized path (product code, torch CPU ops only).

Mirrors, with the same names and argument meaning:
  * fuse_conv_and_bn ............ utils/bn_fuse.py:21-45 (and conv+bn2conv.py:126-150)
  * quantize_tensor(_b) ......... retune_bias_quantize.py:73-97
  * init_quantize_net / quantize_layers ... retune_bias_quantize.py:99-119,
    retune variant retune_bias_quantize_findbest.py:115-148
  * RETUNE table ................ retune_bias_quantize_findbest.py:122-141 == yolo_forward.c:35
Everything here is one-off CPU work on <2M parameters; the hot path never calls it.
"""
import numpy as np
import torch
import torch.nn as nn

RETUNE = [11, 10, 10, 11, 11, 10, 11, 11, 11, 10]


def fuse_conv_and_bn(conv, bn, corrected=False):
    """Fold BatchNorm into the conv (utils/bn_fuse.py:21-45).  The reference leaves
    conv.bias unscaled by gamma/sqrt(var+eps) (bn_fuse.py:42-43, exact only when
    conv.bias == 0); that formula is the default so folded checkpoints match the
    reference's; corrected=True applies the exact fold."""
    with torch.no_grad():
        fusedconv = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size,
                              stride=conv.stride, padding=conv.padding, bias=True)
        w_conv = conv.weight.clone().view(conv.out_channels, -1)
        w_bn = torch.diag(bn.weight.div(torch.sqrt(bn.eps + bn.running_var)))
        fusedconv.weight.copy_(torch.mm(w_bn, w_conv).view(fusedconv.weight.size()))
        b_conv = conv.bias if conv.bias is not None else torch.zeros(conv.weight.size(0))
        if corrected:
            b_conv = b_conv * bn.weight.div(torch.sqrt(bn.running_var + bn.eps))
        b_bn = bn.bias - bn.weight.mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps))
        fusedconv.bias.copy_(b_conv + b_bn)
        return fusedconv


def quantize_tensor(tensor, bitwidth=8, channel_level=False):
    """retune_bias_quantize.py:73-86: (round(scale*t), scale) with scale = 2^floor(log2(127/max|t|))."""
    if channel_level:
        raise NotImplementedError("the reference only uses channel_level=False (retune_bias_quantize.py:115)")
    _max = tensor.abs().max()
    scale = (2 ** (bitwidth - 1) - 1) / _max
    scale = 2 ** torch.floor(torch.log2(scale))
    return torch.round(scale * tensor), scale


quantize_tensor_b = quantize_tensor    # retune_bias_quantize.py:88-97 is the same arithmetic


def to_int8_pow2(t):
    """(q int32 ndarray, e) such that q / 2^e reproduces the reference's stored tensor."""
    q, scale = quantize_tensor(t.detach().float().cpu())
    e = int(torch.log2(scale).item())
    return q.to(torch.int32).numpy(), e


def as_dyadic_int8(t):
    """Interpret an already fake-quantized tensor (values q / 2^e, |q| <= 127) -- the layout
    quantize_layers leaves in the checkpoint.  Raises ValueError if it is not one."""
    q, e = to_int8_pow2(t)
    back = torch.from_numpy(q.astype(np.float32)) * (2.0 ** (-e))
    if not torch.equal(back, t.detach().float().cpu()):
        raise ValueError("tensor is not an 8-bit power-of-two quantized tensor; run "
                         "yolo355.prep.quantize_layers(...) on the model first")
    return q, e


_quantized_layers = []


def init_quantize_net(net, weight_bitwidth=8):
    """retune_bias_quantize.py:99-109: remember fp32 copies of every conv's weight/bias."""
    _quantized_layers.clear()
    for _name, m in net.named_modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            m.weight.weight_back = m.weight.data.clone()
            m.bias.bias_back = m.bias.data.clone()
            _quantized_layers.append(m)
    return _quantized_layers


def quantize_layers(bitwidth=8, rescale=True, retune=False):
    """retune_bias_quantize.py:111-119; retune=True = retune_bias_quantize_findbest.py:115-148
    (weights and biases additionally multiplied by 2^retune[layer])."""
    for i, layer in enumerate(_quantized_layers):
        with torch.no_grad():
            qw, sw = quantize_tensor(layer.weight.weight_back, bitwidth)
            qb, sb = quantize_tensor_b(layer.bias.bias_back, bitwidth)
            r = float(2 ** RETUNE[i]) if (retune and i < len(RETUNE)) else 1.0
            layer.weight[...] = qw * r / sw if rescale else qw
            layer.bias[...] = qb * r / sb if rescale else qb


def quantize_folded(folded):
    """[(W fp32, b fp32)] of BN-folded convs -> [{q_w, e_w, q_b, e_b}]: the per-tensor power-of-two
    int8 recipe of quantize_layers (retune_bias_quantize.py:111-119) for the y355_net graphs."""
    out = []
    for w, b in folded:
        qw, ew = to_int8_pow2(torch.as_tensor(w))
        qb, eb = to_int8_pow2(torch.as_tensor(b))
        out.append(dict(q_w=qw, e_w=ew, q_b=qb, e_b=eb))
    return out


class RangeTracker:
    """Host mirror of AveragedRangeTracker's state (models/slim_yolo_v2.py:9-38) driven by the
    max|activation| the GPU reports.  scale / first_a are the checkpoint buffers."""

    def __init__(self, scale=None, first_a=0, momentum=0.1):
        self.momentum = momentum
        self.scale = torch.zeros(1) if scale is None else scale.detach().clone().float().cpu().reshape(1)
        self.first_a = int(first_a)

    def update(self, max_abs, freeze):
        m = torch.as_tensor(max_abs, dtype=torch.float32).reshape(())
        s = (2 ** (8 - 1) - 1) / m
        if self.first_a == 0:
            self.first_a = 1
            self.scale = self.scale + s
        elif freeze:
            pass
        else:
            self.scale = self.scale * (1 - self.momentum) + s * self.momentum
        return self.exponent()

    def exponent(self):
        return int(torch.floor(torch.log2(self.scale)).item())
