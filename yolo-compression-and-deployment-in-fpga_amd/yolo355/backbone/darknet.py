"""Parameter layout of the reference's backbones (backbone/darknet.py) so that its checkpoints load
unchanged.  DarkNet_Light holds weights only: YOLOv3tiny folds BN and runs the whole graph through the
engine (csrc/net.hip, Y355_ARCH_TINY_V3).  Conv_BN_LeakyReLU, resblock and DarkNet_53 also run stand-alone,
layer by layer, through the operator API (y355_conv2d_bf16: stride-2 convolutions, 1x1 convolutions and the
residual add fused into the convolution's epilogue) -- the building blocks of yolo_v3 / yolo_v3_spp
(SURVEY.md 8f-3); every call is a host round trip, so this form is for parity and bring-up, not for speed."""
import torch.nn as nn

from ..utils.modules import _conv_bn_act_forward


class Conv_BN_LeakyReLU(nn.Module):
    """conv + BatchNorm + LeakyReLU(0.1) (backbone/darknet.py:12-22)."""

    def __init__(self, in_channels, out_channels, ksize, padding=0, stride=1, dilation=1):
        super().__init__()
        self.convs = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, ksize, padding=padding, stride=stride, dilation=dilation),
            nn.BatchNorm2d(out_channels),
            nn.LeakyReLU(0.1, inplace=True))

    def forward(self, x, residual=None):
        return _conv_bn_act_forward(self.convs, x, residual)


class resblock(nn.Module):
    """backbone/darknet.py:24-38: x = module(x) + x, module = 1x1 (ch -> ch/2) then 3x3 (ch/2 -> ch);
    the add rides in the 3x3 convolution's epilogue."""

    def __init__(self, ch, nblocks=1):
        super().__init__()
        self.module_list = nn.ModuleList()
        for _ in range(nblocks):
            self.module_list.append(nn.Sequential(Conv_BN_LeakyReLU(ch, ch // 2, 1),
                                                  Conv_BN_LeakyReLU(ch // 2, ch, 3, padding=1)))

    def forward(self, x):
        for module in self.module_list:
            x = module[1](module[0](x), residual=x)
        return x


class DarkNet_53(nn.Module):
    """backbone/darknet.py:112-161: returns (C_3, C_4, C_5) at strides 8, 16, 32."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.layer_1 = nn.Sequential(Conv_BN_LeakyReLU(3, 32, 3, padding=1),
                                     Conv_BN_LeakyReLU(32, 64, 3, padding=1, stride=2), resblock(64, nblocks=1))
        self.layer_2 = nn.Sequential(Conv_BN_LeakyReLU(64, 128, 3, padding=1, stride=2), resblock(128, nblocks=2))
        self.layer_3 = nn.Sequential(Conv_BN_LeakyReLU(128, 256, 3, padding=1, stride=2), resblock(256, nblocks=8))
        self.layer_4 = nn.Sequential(Conv_BN_LeakyReLU(256, 512, 3, padding=1, stride=2), resblock(512, nblocks=8))
        self.layer_5 = nn.Sequential(Conv_BN_LeakyReLU(512, 1024, 3, padding=1, stride=2), resblock(1024, nblocks=4))

    def forward(self, x, targets=None):
        x = self.layer_1(x)
        x = self.layer_2(x)
        c3 = self.layer_3(x)
        c4 = self.layer_4(c3)
        c5 = self.layer_5(c4)
        return c3, c4, c5


class DarkNet_Light(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv_1 = Conv_BN_LeakyReLU(3, 16, 3, 1)
        self.maxpool_1 = nn.MaxPool2d((2, 2), 2)
        self.conv_2 = Conv_BN_LeakyReLU(16, 32, 3, 1)
        self.maxpool_2 = nn.MaxPool2d((2, 2), 2)
        self.conv_3 = Conv_BN_LeakyReLU(32, 64, 3, 1)
        self.maxpool_3 = nn.MaxPool2d((2, 2), 2)
        self.conv_4 = Conv_BN_LeakyReLU(64, 128, 3, 1)
        self.maxpool_4 = nn.MaxPool2d((2, 2), 2)
        self.conv_5 = Conv_BN_LeakyReLU(128, 256, 3, 1)
        self.maxpool_5 = nn.MaxPool2d((2, 2), 2)
        self.conv_6 = Conv_BN_LeakyReLU(256, 512, 3, 1)
        self.maxpool_6 = nn.Sequential(nn.ZeroPad2d((0, 1, 0, 1)), nn.MaxPool2d((2, 2), 1))
        self.conv_7 = Conv_BN_LeakyReLU(512, 1024, 3, 1)

    def forward(self, x):
        raise NotImplementedError("yolo355: the backbone runs inside YOLOv3tiny's engine graph")


def darknet53(pretrained=False, hr=False, **kwargs):
    """backbone/darknet.py:273-288; pretrained ImageNet weights are a file the caller loads."""
    if pretrained:
        raise NotImplementedError("yolo355: load pretrained backbone weights with load_state_dict")
    return DarkNet_53()


def darknet_light(pretrained=False, hr=False, **kwargs):
    """backbone/darknet.py:295-310; pretrained ImageNet weights are a file the caller loads."""
    if pretrained:
        raise NotImplementedError("yolo355: load pretrained backbone weights with load_state_dict")
    return DarkNet_Light()


class _Pool2(nn.Module):
    """nn.MaxPool2d((2, 2), 2) through y355_maxpool2x2_f32 (bit-exact)."""

    def forward(self, x):
        import torch
        from ..engine import maxpool2x2_f32
        y = maxpool2x2_f32(x.detach().float().cpu().numpy(), device_id=x.device.index if x.is_cuda and x.device.index is not None else 0)
        return torch.from_numpy(y).to(x.device)


class DarkNet_19(nn.Module):
    """backbone/darknet.py:40-110: returns (C_4, C_5, C_6) at strides 8, 16, 32.  The pools are parameter-free, so the
    state_dict layout equals the reference's (conv_k.N.convs.*)."""

    def __init__(self, num_classes=1000):
        super().__init__()
        C = Conv_BN_LeakyReLU
        self.conv_1 = nn.Sequential(C(3, 32, 3, 1), _Pool2())
        self.conv_2 = nn.Sequential(C(32, 64, 3, 1), _Pool2())
        self.conv_3 = nn.Sequential(C(64, 128, 3, 1), C(128, 64, 1), C(64, 128, 3, 1), _Pool2())
        self.conv_4 = nn.Sequential(C(128, 256, 3, 1), C(256, 128, 1), C(128, 256, 3, 1))
        self.maxpool_4 = _Pool2()
        self.conv_5 = nn.Sequential(C(256, 512, 3, 1), C(512, 256, 1), C(256, 512, 3, 1), C(512, 256, 1), C(256, 512, 3, 1))
        self.maxpool_5 = _Pool2()
        self.conv_6 = nn.Sequential(C(512, 1024, 3, 1), C(1024, 512, 1), C(512, 1024, 3, 1), C(1024, 512, 1), C(512, 1024, 3, 1))

    def forward(self, x):
        x = self.conv_1(x)
        x = self.conv_2(x)
        x = self.conv_3(x)
        c4 = self.conv_4(x)
        c5 = self.conv_5(self.maxpool_4(c4))
        c6 = self.conv_6(self.maxpool_5(c5))
        return c4, c5, c6


def darknet19(pretrained=False, hr=False, **kwargs):
    """backbone/darknet.py:257-271; pretrained ImageNet weights are a file the caller loads."""
    if pretrained:
        raise NotImplementedError("yolo355: load pretrained backbone weights with load_state_dict")
    return DarkNet_19()
