"""Drop-in for the reference's models/tiny_yolo_v3.py (YOLOv3tiny, :9-273): same constructor,
module names / state_dict keys and eval-mode return, backed by the MI355X engine
(include/yolo355.h, y355_net with Y355_ARCH_TINY_V3).

    net = YOLOv3tiny(device, input_size=[416, 416], num_classes=20, anchor_size=TINY_MULTI_ANCHOR_SIZE)
    net.load_state_dict(torch.load(...)); net.eval()
    bboxes, scores, cls_inds = net(x)            # numpy, image 0, anchor order (stride 16 level first)
    all_images = net.forward_batch(x)            # new: every image of the batch
    bboxes, scores, cls_inds = net(x, quantization=True)   # new: int8 engine (power-of-two PTQ)
"""
import numpy as np
import torch
import torch.nn as nn

from ..backbone import darknet_light
from ..utils import Conv2d
from .slim_yolo_v2 import _NetModel


class YOLOv3tiny(_NetModel):
    _arch = "tiny_yolo_v3"

    def __init__(self, device, input_size=None, num_classes=20, trainable=False, conf_thresh=0.01,
                 nms_thresh=0.50, anchor_size=None, hr=False):
        super().__init__()
        self.device = device
        self.input_size = list(input_size)
        self.num_classes = num_classes
        self.trainable = trainable
        self.conf_thresh = conf_thresh
        self.nms_thresh = nms_thresh
        self.stride = [16, 32]
        self.anchor_size = torch.tensor(anchor_size).view(2, len(anchor_size) // 2, 2)
        self.anchor_number = self.anchor_size.size(1)
        self.scale = np.array([[[input_size[1], input_size[0], input_size[1], input_size[0]]]])
        self.backbone = darknet_light(pretrained=False, hr=hr)
        self.conv_set_2 = Conv2d(1024, 256, 3, padding=1, leakyReLU=True)
        self.conv_1x1_2 = Conv2d(256, 128, 1, leakyReLU=True)
        self.extra_conv_2 = Conv2d(256, 512, 3, padding=1, leakyReLU=True)
        self.pred_2 = nn.Conv2d(512, self.anchor_number * (1 + 4 + self.num_classes), 1)
        self.conv_set_1 = Conv2d(384, 256, 3, padding=1, leakyReLU=True)
        self.pred_1 = nn.Conv2d(256, self.anchor_number * (1 + 4 + self.num_classes), 1)

    def _conv_modules(self):
        """weight slots of csrc/net.hip, forward order (tiny_yolo_v3.py:176-200)."""
        bb = self.backbone
        return [bb.conv_1.convs, bb.conv_2.convs, bb.conv_3.convs, bb.conv_4.convs, bb.conv_5.convs,
                bb.conv_6.convs, bb.conv_7.convs, self.conv_set_2.convs, self.conv_1x1_2.convs,
                self.conv_set_1.convs, self.extra_conv_2.convs, self.pred_2, self.pred_1]

    def forward(self, x, target=None, quantization=False):
        """Eval-mode return of the reference (:224-243): detections of image 0.  quantization=True is
        this build's int8 form of the model (the reference has none): see _NetModel.forward_batch."""
        return self.forward_batch(x, quantization=quantization)[0]
