"""Drop-ins for the reference's `myYOLOv3` (models/yolo_v3.py:9-304) and `myYOLOv3Spp` (models/yolo_v3_spp.py):
DarkNet-53 backbone (stride-2 convolutions, residual blocks), three prediction levels (strides 8, 16, 32) joined by
1x1 convolutions and bilinear x2 up-sampling, optional SPP in front of the stride-32 branch.  Same constructor,
attribute names (checkpoints load unchanged) and eval-mode return value.  `forward` / `forward_batch` run the whole graph
on the GPU through y355_net (Y355_ARCH_YOLO_V3 / _SPP: 75 BN-folded convolutions on the bf16 MFMA, stride-2 and residual
forms of the generic kernel, SPP and bilinear x2 into concat buffers, three-level head -- csrc/net.hip).
`forward_batch_composed` / `prediction_maps` run the same graph layer by layer through the operator API
(y355_conv2d_bf16, y355_spp_f32, y355_upsample2x_f32, y355_head_f32 -- SURVEY.md 8f-3): the bring-up / cross-check form.  At 416 x 416 an
image has 10 647 anchors: the head thresholds and compacts them on the GPU, and at most 4096 may pass conf_thresh.
Training is not built."""
import numpy as np
import torch
import torch.nn as nn

from ..backbone.darknet import darknet53
from ..utils.modules import Conv2d, SPP
from .slim_yolo_v2 import _NetModel


def _dev(x):
    return x.device.index if x.is_cuda and x.device.index is not None else 0


def _pred(conv, x):
    from ..engine import conv2d_bf16
    return conv2d_bf16(x.detach().float().cpu().numpy(), conv.weight.detach().float().cpu().numpy(),
                       conv.bias.detach().float().cpu().numpy(), stride=1, neg_slope=1.0, out_fp32=True, device_id=_dev(x))


def _up(x):
    from ..engine import upsample2x_f32
    return torch.from_numpy(upsample2x_f32(x.detach().float().cpu().numpy(), device_id=_dev(x))).to(x.device)


class myYOLOv3(_NetModel):
    _spp = False
    _arch = "yolo_v3"

    def __init__(self, device, input_size=None, num_classes=20, trainable=False, conf_thresh=0.001, nms_thresh=0.50,
                 anchor_size=None, hr=False):
        super().__init__()
        if trainable:
            raise NotImplementedError("yolo355 is an inference engine: trainable=True is not built")
        self.device = device
        self.input_size = list(input_size)
        self.num_classes = num_classes
        self.trainable = trainable
        self.conf_thresh = conf_thresh
        self.nms_thresh = nms_thresh
        self.stride = [8, 16, 32]
        self.anchor_size = torch.tensor(anchor_size).view(3, len(anchor_size) // 3, 2)
        self.anchor_number = self.anchor_size.size(1)
        self.scale = np.array([[[input_size[1], input_size[0], input_size[1], input_size[0]]]])
        self.backbone = darknet53(pretrained=False, hr=hr)
        A, C = self.anchor_number, self.num_classes
        first = [SPP(), Conv2d(1024 * 4, 512, 1, leakyReLU=True)] if self._spp else [Conv2d(1024, 512, 1, leakyReLU=True)]
        self.conv_set_3 = nn.Sequential(*first, Conv2d(512, 1024, 3, padding=1, leakyReLU=True),
                                        Conv2d(1024, 512, 1, leakyReLU=True), Conv2d(512, 1024, 3, padding=1, leakyReLU=True),
                                        Conv2d(1024, 512, 1, leakyReLU=True))
        self.conv_1x1_3 = Conv2d(512, 256, 1, leakyReLU=True)
        self.extra_conv_3 = Conv2d(512, 1024, 3, padding=1, leakyReLU=True)
        self.pred_3 = nn.Conv2d(1024, A * (1 + 4 + C), 1)
        self.conv_set_2 = nn.Sequential(Conv2d(768, 256, 1, leakyReLU=True), Conv2d(256, 512, 3, padding=1, leakyReLU=True),
                                        Conv2d(512, 256, 1, leakyReLU=True), Conv2d(256, 512, 3, padding=1, leakyReLU=True),
                                        Conv2d(512, 256, 1, leakyReLU=True))
        self.conv_1x1_2 = Conv2d(256, 128, 1, leakyReLU=True)
        self.extra_conv_2 = Conv2d(256, 512, 3, padding=1, leakyReLU=True)
        self.pred_2 = nn.Conv2d(512, A * (1 + 4 + C), 1)
        self.conv_set_1 = nn.Sequential(Conv2d(384, 128, 1, leakyReLU=True), Conv2d(128, 256, 3, padding=1, leakyReLU=True),
                                        Conv2d(256, 128, 1, leakyReLU=True), Conv2d(128, 256, 3, padding=1, leakyReLU=True),
                                        Conv2d(256, 128, 1, leakyReLU=True))
        self.extra_conv_1 = Conv2d(128, 256, 3, padding=1, leakyReLU=True)
        self.pred_1 = nn.Conv2d(256, A * (1 + 4 + C), 1)

    def _conv_modules(self):
        """weight slots of csrc/net.hip (V3Graph), forward order: DarkNet-53 (52), conv_set_3, conv_1x1_3, conv_set_2,
        conv_1x1_2, conv_set_1, then extra_conv / pred of strides 32, 16, 8"""
        bb = self.backbone
        mods = []
        for layer in (bb.layer_1, bb.layer_2, bb.layer_3, bb.layer_4, bb.layer_5):
            for m in layer:
                if hasattr(m, "module_list"):
                    for blk in m.module_list:
                        mods += [blk[0].convs, blk[1].convs]
                else:
                    mods.append(m.convs)
        mods += [m.convs for m in self.conv_set_3 if hasattr(m, "convs")]
        mods.append(self.conv_1x1_3.convs)
        mods += [m.convs for m in self.conv_set_2]
        mods.append(self.conv_1x1_2.convs)
        mods += [m.convs for m in self.conv_set_1]
        mods += [self.extra_conv_3.convs, self.pred_3, self.extra_conv_2.convs, self.pred_2, self.extra_conv_1.convs, self.pred_1]
        return mods

    def prediction_maps(self, x):
        """[pred_1 (stride 8), pred_2 (16), pred_3 (32)], each [B, A*(5+C), H/s, W/s] fp32 (models/yolo_v3.py:203-231)."""
        fmp_1, fmp_2, fmp_3 = self.backbone(x)
        fmp_3 = self.conv_set_3(fmp_3)
        fmp_3_up = _up(self.conv_1x1_3(fmp_3))
        fmp_2 = self.conv_set_2(torch.cat([fmp_2, fmp_3_up], 1))
        fmp_2_up = _up(self.conv_1x1_2(fmp_2))
        fmp_1 = self.conv_set_1(torch.cat([fmp_1, fmp_2_up], 1))
        p3 = _pred(self.pred_3, self.extra_conv_3(fmp_3))
        p2 = _pred(self.pred_2, self.extra_conv_2(fmp_2))
        p1 = _pred(self.pred_1, self.extra_conv_1(fmp_1))
        return [p1, p2, p3]

    def forward_batch_composed(self, x):
        from ..engine import head_f32
        if self.training:
            raise NotImplementedError("yolo355 is an inference engine: call .eval() first")
        with torch.no_grad():
            preds = self.prediction_maps(x)
        return head_f32(preds, self.stride, self.anchor_size.detach().float().cpu().numpy(), self.num_classes, self.input_size,
                        1.0, self.conf_thresh, self.nms_thresh, device_id=_dev(x))

    def forward_batch(self, x, quantization=False):
        if quantization:
            raise NotImplementedError("yolo355: yolo_v3 has no quantized form (neither has the reference)")
        return super().forward_batch(x)

    def forward(self, x, target=None):
        if target is not None:
            raise NotImplementedError("yolo355 is an inference engine: the training branch is not built")
        return self.forward_batch(x)[0]          # the reference decodes batch element 0 only (:267-269)


class myYOLOv3Spp(myYOLOv3):
    """models/yolo_v3_spp.py: SPP + Conv2d(4096, 512, 1) open the stride-32 branch."""
    _spp = True
    _arch = "yolo_v3_spp"
