"""Drop-in for the reference's `myYOLOv2` (models/yolo_v2.py:9-232): DarkNet-19 backbone, reorg route, one
prediction level at stride 32.  Same constructor, attribute names (so checkpoints load unchanged) and eval-mode
return value.  `forward` / `forward_batch` run the whole graph on the GPU through y355_net (Y355_ARCH_YOLO_V2: 23
BN-folded convolutions on the bf16 MFMA, pools fused or stand-alone, reorg + concat by channel offsets, head + NMS
-- csrc/net.hip).  `forward_batch_composed` / `prediction_map` run the same graph layer by layer through the
operator API (y355_conv2d_bf16, y355_maxpool2x2_f32, y355_reorg_f32, y355_head_f32 -- SURVEY.md 8f-3), one host round
trip per layer: the bring-up / cross-check form.  Training is not built."""
import numpy as np
import torch
import torch.nn as nn

from ..backbone.darknet import darknet19
from ..utils.modules import Conv2d, reorg_layer, _conv_bn_act_forward
from .slim_yolo_v2 import _NetModel


class myYOLOv2(_NetModel):
    _arch = "yolo_v2"

    def __init__(self, device, input_size=None, num_classes=20, trainable=False, conf_thresh=0.001, nms_thresh=0.5,
                 anchor_size=None, hr=False):
        super().__init__()
        if trainable:
            raise NotImplementedError("yolo355 is an inference engine: myYOLOv2(trainable=True) is not built")
        self.device = device
        self.input_size = list(input_size)
        self.num_classes = num_classes
        self.trainable = trainable
        self.conf_thresh = conf_thresh
        self.nms_thresh = nms_thresh
        self.anchor_size = torch.tensor(anchor_size)
        self.anchor_number = len(anchor_size)
        self.stride = 32
        self.scale = np.array([[[input_size[1], input_size[0], input_size[1], input_size[0]]]])
        self.backbone = darknet19(pretrained=False, hr=hr)
        self.convsets_1 = nn.Sequential(Conv2d(1024, 1024, 3, 1, leakyReLU=True), Conv2d(1024, 1024, 3, 1, leakyReLU=True))
        self.route_layer = Conv2d(512, 64, 1, leakyReLU=True)
        self.reorg = reorg_layer(stride=2)
        self.convsets_2 = Conv2d(1280, 1024, 3, 1, leakyReLU=True)
        self.pred = nn.Conv2d(1024, self.anchor_number * (1 + 4 + self.num_classes), 1)

    def _conv_modules(self):
        """weight slots of csrc/net.hip (kV2Ops), forward order of models/yolo_v2.py:165-179"""
        bb = self.backbone
        mods = [bb.conv_1[0], bb.conv_2[0]] + [bb.conv_3[i] for i in range(3)] + [bb.conv_4[i] for i in range(3)] + \
               [bb.conv_5[i] for i in range(5)] + [bb.conv_6[i] for i in range(5)] + \
               [self.convsets_1[0], self.convsets_1[1], self.route_layer, self.convsets_2]
        return [m.convs for m in mods] + [self.pred]

    def prediction_map(self, x):
        """[B, A*(5+C), H/32, W/32] fp32 (models/yolo_v2.py:165-179)."""
        _, fp_1, fp_2 = self.backbone(x)
        fp_2 = self.convsets_1(fp_2)
        fp_1 = self.reorg(self.route_layer(fp_1))
        fp = torch.cat([fp_1, fp_2], dim=1)
        fp = self.convsets_2(fp)
        from ..engine import conv2d_bf16
        w = self.pred.weight.detach().float().cpu().numpy()
        b = self.pred.bias.detach().float().cpu().numpy()
        y = conv2d_bf16(fp.detach().float().cpu().numpy(), w, b, stride=1, neg_slope=1.0, out_fp32=True,
                        device_id=x.device.index if x.is_cuda and x.device.index is not None else 0)
        return y

    def forward_batch_composed(self, x):
        from ..engine import head_f32
        if self.training:
            raise NotImplementedError("yolo355 is an inference engine: call .eval() first")
        with torch.no_grad():
            pred = self.prediction_map(x)
        anchors = self.anchor_size.detach().float().cpu().numpy().reshape(1, -1, 2)
        return head_f32([pred], [self.stride], anchors, self.num_classes, self.input_size, float(self.stride),
                        self.conf_thresh, self.nms_thresh,
                        device_id=x.device.index if x.is_cuda and x.device.index is not None else 0)

    def forward_batch(self, x, quantization=False):
        if quantization:
            raise NotImplementedError("yolo355: yolo_v2 has no quantized form (neither has the reference)")
        return super().forward_batch(x)

    def forward(self, x, target=None):
        if target is not None:
            raise NotImplementedError("yolo355 is an inference engine: the training branch (models/yolo_v2.py:212-232) is not built")
        return self.forward_batch(x)[0]          # the reference decodes batch element 0 only (:200-204)
