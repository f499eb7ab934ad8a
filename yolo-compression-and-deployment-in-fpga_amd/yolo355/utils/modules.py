"""Drop-in operator modules with the reference's signatures (utils/modules.py:6-40).

The classes keep the reference's parameter layout (`.convs` = nn.Sequential(nn.Conv2d, [BN],
activation)) so checkpoints load unchanged.  `forward` of the BN-folded modules runs on the
MI355X through the C ABI when the operands are the fake-quantized tensors of the quantized
path (values q / 2^e with |q| <= 127, the only regime the FPGA path and this engine define):
the result is the exact fp32 tensor the reference's nn.Conv2d + LeakyReLU(0.125) produces.
Operands that are NOT dyadic int8 values (utils/modules.py:28-29 accepts any fp32 tensor) run the same layer on the bf16
MFMA through y355_conv2d_bf16 (operands and result rounded to bf16, the tolerance of the fp32 model families); training
raises -- there is no CPU / PyTorch fallback.

Round 6: a CUDA tensor stays on the GPU.  The modules keep their weights packed on the device (engine.ConvOp = y355_conv_op,
re-packed when a parameter's version changes) and call the device-pointer entry points on torch's current stream; the result is
a CUDA tensor and no tensor crosses to the host.  (Conv2d_fuse's int8 route reads back eight bytes: the input's exponent and
the verdict whether it is a dyadic int8 tensor decide the route on the host.)  CPU tensors take the host-pointer entry points.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import prep


def _versions(convs):
    mods = list(convs) if isinstance(convs, nn.Sequential) else [convs]
    t = [p for m in mods for p in list(m.parameters()) + list(m.buffers())]
    return tuple(int(p._version) for p in t) + tuple(p.data_ptr() for p in t)


def _cached_op(owner, key, convs, build):
    """the owner module's device-resident operator for `key`, rebuilt when a parameter or buffer of `convs` changed"""
    cache = owner.__dict__.setdefault("_y355_ops", {})
    ver = _versions(convs)
    hit = cache.get(key)
    if hit is None or hit[0] != ver:
        if hit is not None:
            hit[1].close()
        cache[key] = (ver, build())
    return cache[key][1]


def _int8_operands(module, x):
    conv = module.convs[0]
    if conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.padding != (1, 1) or conv.dilation != (1, 1):
        raise NotImplementedError("yolo355 fused conv: only 3x3 / stride 1 / pad 1 (the slim-YOLOv2 layers)")
    try:
        q_in, sa_in = prep.as_dyadic_int8(x)
        q_w, e_w = prep.as_dyadic_int8(conv.weight)
        if conv.bias is not None:
            q_b, e_b = prep.as_dyadic_int8(conv.bias)
        else:
            q_b, e_b = np.zeros(conv.out_channels, np.int32), 0
    except ValueError:
        return None                                       # not fake-quantized operands: the bf16 route
    return q_in, sa_in, q_w, e_w, q_b, e_b


class _NoOp:
    def close(self):
        pass


class _FusedBase(nn.Module):
    leaky = True

    def forward(self, x):
        from ..engine import ConvOp, conv3x3_i8_raw
        if x.is_cuda:
            conv = self.convs[0]
            if conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.padding != (1, 1) or conv.dilation != (1, 1):
                raise NotImplementedError("yolo355 fused conv: only 3x3 / stride 1 / pad 1 (the slim-YOLOv2 layers)")

            def build():
                try:
                    q_w, e_w = prep.as_dyadic_int8(conv.weight)
                    q_b, e_b = prep.as_dyadic_int8(conv.bias) if conv.bias is not None else (np.zeros(conv.out_channels, np.int32), 0)
                except ValueError:
                    return _NoOp()                        # weights that are not dyadic: always the bf16 route
                return ConvOp.int8(q_w, q_b, e_w, e_b, leaky=self.leaky, relu=not self.leaky, device=x.device)
            op = _cached_op(self, ("i8", x.device.index), self.convs, build)
            y = op.forward_i8(x) if not isinstance(op, _NoOp) else None
            return y if y is not None else _conv_bn_act_forward(self.convs, x)
        ops = _int8_operands(self, x)
        if ops is None:
            # the reference's module takes any fp32 tensor (utils/modules.py:28-29, :39-40): same layer on the bf16 MFMA
            return _conv_bn_act_forward(self.convs, x)
        q_in, sa_in, q_w, e_w, q_b, e_b = ops
        t, frac = conv3x3_i8_raw(q_in, q_w, q_b, sa_in, e_w, e_b, leaky=self.leaky, relu=not self.leaky,
                                 device_id=x.device.index if x.is_cuda and x.device.index is not None else 0)
        y = torch.from_numpy(t.astype(np.float32) * np.float32(2.0 ** (-frac)))
        return y.to(x.device)


def folded_f32(convs):
    """Exact eval-mode fold of nn.Sequential(conv, BatchNorm2d, act) (or a bare conv) into fp32
    (W', b') numpy arrays: W' = W * g, b' = (b - mean) * g + beta, g = gamma / sqrt(var + eps).
    (The reference's fuse_conv_and_bn, utils/bn_fuse.py:42-43, leaves b unscaled; the fp32 model's
    own forward is conv -> BN, which this fold reproduces.)"""
    conv = convs[0] if isinstance(convs, nn.Sequential) else convs
    w = conv.weight.detach().double().cpu()
    b = conv.bias.detach().double().cpu() if conv.bias is not None else torch.zeros(w.shape[0], dtype=torch.float64)
    if isinstance(convs, nn.Sequential) and len(convs) > 1 and isinstance(convs[1], nn.BatchNorm2d):
        bn = convs[1]
        g = bn.weight.detach().double().cpu() / torch.sqrt(bn.running_var.detach().double().cpu() + bn.eps)
        w = w * g.view(-1, 1, 1, 1)
        b = (b - bn.running_mean.detach().double().cpu()) * g + bn.bias.detach().double().cpu()
    return w.float().numpy(), b.float().numpy()


def _conv_bn_act_forward(convs, x, residual=None):
    """Eval-mode forward of nn.Sequential(conv, [BatchNorm2d], [LeakyReLU | ReLU]) through y355_conv2d_bf16."""
    from ..engine import conv2d_bf16
    conv = convs[0]
    k = conv.kernel_size[0]
    if conv.kernel_size[0] != conv.kernel_size[1] or k not in (1, 3) or conv.padding != (k // 2, k // 2) \
            or conv.dilation != (1, 1) or conv.groups != 1 or conv.stride[0] != conv.stride[1]:
        raise NotImplementedError("yolo355 conv: 1x1, or 3x3 with padding 1; stride 1 (or 2 for 3x3); no dilation / groups")
    if any(isinstance(m, nn.BatchNorm2d) and m.training for m in convs):
        raise NotImplementedError("yolo355 is an inference engine: call .eval() first (BatchNorm uses running statistics)")
    slope = 1.0
    for m in convs:
        if isinstance(m, nn.LeakyReLU):
            slope = float(m.negative_slope)
        elif isinstance(m, nn.ReLU):
            slope = 0.0
    if x.is_cuda:                                         # device-resident: weights packed once, the tensor never leaves the GPU
        from ..engine import ConvOp
        op = _cached_op(convs, ("bf16", x.device.index), convs,
                        lambda: ConvOp.bf16(*folded_f32(convs), stride=conv.stride[0], neg_slope=slope, device=x.device))
        return op.forward(x, residual)
    w, b = folded_f32(convs)
    dev = x.device
    y = conv2d_bf16(x.detach().float().cpu().numpy(), w, b,
                    None if residual is None else residual.detach().float().cpu().numpy(),
                    stride=conv.stride[0], neg_slope=slope, device_id=dev.index if x.is_cuda and dev.index is not None else 0)
    return torch.from_numpy(y).to(dev)


class Conv2d(nn.Module):
    """conv + BatchNorm + LeakyReLU(0.125)/ReLU (utils/modules.py:6-18): the building block of the
    fp32 models (SlimYOLOv2, YOLOv2/v3 heads).  The whole-model classes fold BN at load time and run
    their graphs through y355_net; a stand-alone call folds BN (eval mode) and runs this one layer on
    the bf16 MFMA through y355_conv2d_bf16 -- operands and result rounded to bf16."""

    def __init__(self, in_channels, out_channels, ksize, padding=0, stride=1, dilation=1, leakyReLU=False):
        super().__init__()
        self.convs = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, ksize, stride=stride, padding=padding, dilation=dilation),
            nn.BatchNorm2d(out_channels),
            nn.LeakyReLU(0.125, inplace=True) if leakyReLU else nn.ReLU(inplace=True))

    def forward(self, x):
        return _conv_bn_act_forward(self.convs, x)


class Conv2d_fuse(_FusedBase):
    """conv(+bias) + LeakyReLU(0.125)/ReLU (utils/modules.py:20-29)."""

    def __init__(self, in_channels, out_channels, ksize, padding=0, stride=1, dilation=1, leakyReLU=False):
        super().__init__()
        self.leaky = bool(leakyReLU)
        self.convs = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, ksize, stride=stride, padding=padding, dilation=dilation),
            nn.LeakyReLU(0.125, inplace=True) if leakyReLU else nn.ReLU(inplace=True))


class Conv2d_fuse_nobias(_FusedBase):
    """conv (no bias) + LeakyReLU(0.125)/ReLU (utils/modules.py:31-40)."""

    def __init__(self, in_channels, out_channels, ksize, padding=0, stride=1, dilation=1, leakyReLU=False):
        super().__init__()
        self.leaky = bool(leakyReLU)
        self.convs = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, ksize, stride=stride, padding=padding, dilation=dilation, bias=False),
            nn.LeakyReLU(0.125, inplace=True) if leakyReLU else nn.ReLU(inplace=True))


class reorg_layer(nn.Module):
    """utils/modules.py:43-57: [B, C, H, W] -> [B, C*s*s, H/s, W/s], out channel (sy*s + sx)*C + c (used by yolo_v2).
    Runs y355_reorg_f32: bit-exact."""

    def __init__(self, stride):
        super().__init__()
        self.stride = stride

    def forward(self, x):
        from ..engine import reorg_f32, reorg_f32_dev
        if x.is_cuda:
            return reorg_f32_dev(x, self.stride)
        y = reorg_f32(x.detach().float().cpu().numpy(), self.stride, device_id=x.device.index if x.is_cuda and x.device.index is not None else 0)
        return torch.from_numpy(y).to(x.device)


class SPP(nn.Module):
    """utils/modules.py:59-72: cat(x, max_pool 5, 9, 13 (stride 1, same size)) (used by yolo_v3_spp).
    Runs y355_spp_f32: bit-exact."""

    def forward(self, x):
        from ..engine import spp_f32, spp_f32_dev
        if x.is_cuda:
            return spp_f32_dev(x)
        y = spp_f32(x.detach().float().cpu().numpy(), device_id=x.device.index if x.is_cuda and x.device.index is not None else 0)
        return torch.from_numpy(y).to(x.device)
