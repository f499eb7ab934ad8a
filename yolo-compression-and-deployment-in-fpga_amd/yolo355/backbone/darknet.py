"""Parameter layout of the reference's DarkNet_Light (backbone/darknet.py:211-255) so that its
checkpoints load unchanged.  The modules hold weights only: YOLOv3tiny folds BN and runs the whole
graph through the engine (csrc/net.hip, Y355_ARCH_TINY_V3)."""
import torch.nn as nn


class Conv_BN_LeakyReLU(nn.Module):
    """conv + BatchNorm + LeakyReLU(0.1) (backbone/darknet.py:12-22)."""

    def __init__(self, in_channels, out_channels, ksize, padding=0, stride=1, dilation=1):
        super().__init__()
        self.convs = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, ksize, padding=padding, stride=stride, dilation=dilation),
            nn.BatchNorm2d(out_channels),
            nn.LeakyReLU(0.1, inplace=True))

    def forward(self, x):
        raise NotImplementedError("yolo355: stand-alone backbone blocks are not built; run YOLOv3tiny")


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


def darknet_light(pretrained=False, hr=False, **kwargs):
    """backbone/darknet.py:295-310; pretrained ImageNet weights are a file the caller loads."""
    if pretrained:
        raise NotImplementedError("yolo355: load pretrained backbone weights with load_state_dict")
    return DarkNet_Light()
