"""Drop-in for the reference's models/slim_yolo_v2.py quantized model, backed by the MI355X
engine (include/yolo355.h).

Same constructor, attributes, state_dict layout and eval-mode return as
`SlimYOLOv2_quantize_bnfuse` (models/slim_yolo_v2.py:40-382):

    net = SlimYOLOv2_quantize_bnfuse(device, input_size=[416, 416], num_classes=2,
                                     anchor_size=ANCHOR_SIZE_MASK)
    net.load_state_dict(torch.load(".../slim_yolo_v2_q_bf_retune_quantize_1.pth"), strict=False)
    net.eval()
    bboxes, scores, cls_inds = net(x, quantization=True)      # numpy, image 0, anchor order
    all_images = net.forward_batch(x)                          # new: every image of the batch

What runs where: parameters live in torch modules (checkpoint compatibility only); every
forward goes through the C ABI into the HIP kernels.  There is no PyTorch compute path:
`quantization=True` runs the int8 engine (y355_engine), the default `quantization=False`
(the call of test.py:84 / demo.py:81 / utils/vocapi_evaluator.py:67: trackers are the identity,
fp32 math on whatever weights are loaded) runs the same graph on the bf16 MFMA (y355_net);
`trainable=True` (losses) raises NotImplementedError.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import prep
from ..engine import Engine, Pipeline
from ..netengine import Net
from ..utils import Conv2d, Conv2d_fuse
from ..utils.modules import folded_f32

_CONVS = ["conv1", "conv2", "conv3_1", "conv3_2", "conv4_1", "conv4_2", "conv5", "conv6", "conv7"]
_TRACKERS = ["a_tracker_in", "a_tracker1", "a_tracker2", "a_tracker3_1", "a_tracker3_2", "a_tracker4_1",
             "a_tracker4_2", "a_tracker5", "a_tracker6", "a_tracker7", "a_tracker_pred"]


class AveragedRangeTracker(nn.Module):
    """Checkpoint-compatible holder of the tracker buffers (models/slim_yolo_v2.py:9-14).
    The max|activation| statistics come from the GPU; the state update is prep.RangeTracker."""

    def __init__(self, momentum=0.1):
        super().__init__()
        self.momentum = momentum
        self.register_buffer("scale", torch.zeros(1))
        self.register_buffer("first_a", torch.zeros(1))


class SlimYOLOv2_quantize_bnfuse(nn.Module):
    def __init__(self, device, input_size=None, num_classes=20, trainable=False, conf_thresh=0.01,
                 nms_thresh=0.5, anchor_size=None, hr=False):
        super().__init__()
        self.device = device
        self.input_size = list(input_size)
        self.num_classes = num_classes
        self.trainable = trainable
        self.conf_thresh = conf_thresh
        self.nms_thresh = nms_thresh
        self.anchor_size = torch.tensor(anchor_size)
        self.anchor_number = len(anchor_size)
        self.stride = 16
        self.scale = np.array([[[input_size[1], input_size[0], input_size[1], input_size[0]]]])

        self.a_tracker_in = AveragedRangeTracker()
        self.conv1 = Conv2d_fuse(3, 16, 3, 1, leakyReLU=True)
        self.a_tracker1 = AveragedRangeTracker()
        self.pool1 = nn.MaxPool2d(2, 2)
        self.conv2 = Conv2d_fuse(16, 32, 3, 1, leakyReLU=True)
        self.a_tracker2 = AveragedRangeTracker()
        self.pool2 = nn.MaxPool2d(2, 2)
        self.conv3_1 = Conv2d_fuse(32, 64, 3, 1, leakyReLU=True)
        self.a_tracker3_1 = AveragedRangeTracker()
        self.conv3_2 = Conv2d_fuse(64, 64, 3, 1, leakyReLU=True)
        self.a_tracker3_2 = AveragedRangeTracker()
        self.pool3 = nn.MaxPool2d(2, 2)
        self.conv4_1 = Conv2d_fuse(64, 128, 3, 1, leakyReLU=True)
        self.a_tracker4_1 = AveragedRangeTracker()
        self.conv4_2 = Conv2d_fuse(128, 128, 3, 1, leakyReLU=True)
        self.a_tracker4_2 = AveragedRangeTracker()
        self.pool4 = nn.MaxPool2d(2, 2)
        self.conv5 = Conv2d_fuse(128, 256, 3, 1, leakyReLU=True)
        self.a_tracker5 = AveragedRangeTracker()
        self.conv6 = Conv2d_fuse(256, 256, 3, 1, leakyReLU=True)
        self.a_tracker6 = AveragedRangeTracker()
        self.conv7 = Conv2d_fuse(256, 256, 3, 1, leakyReLU=True)
        self.a_tracker7 = AveragedRangeTracker()
        self.pred = nn.Conv2d(256, self.anchor_number * (1 + 4 + self.num_classes), 3, 1, padding=1)
        self.a_tracker_pred = AveragedRangeTracker()

        self._engine = None
        self._engine_key = None
        self._loaded_version = None
        self._loaded_find = None
        self._pipe = None               # y355_pipeline: batches larger than PIPELINE_CHUNK and the asynchronous submit / collect
        self._pipe_key = None
        self._pipe_loaded = None

    PIPELINE_CHUNK = 64                 # images per ticket of the pipeline (BASELINE.json's batch: the kernels' tuned regime)

    # ------------------------------------------------------------------ reference API
    def set_grid(self, input_size):
        """models/slim_yolo_v2.py:105-109: change the network input size."""
        self.input_size = list(input_size)
        self.scale = np.array([[[input_size[1], input_size[0], input_size[1], input_size[0]]]])

    def forward(self, x, target=None, quantization=False, find=False):
        """Eval-mode return of the reference (:344-358): detections of image 0 as NumPy arrays
        (bboxes float32 [n,4] in [0,1], scores float32 [n], cls_inds int64 [n]), anchor order."""
        return self.forward_batch(x, quantization=quantization, find=find)[0]

    def forward_batch(self, x, quantization=True, find=False, sizes_wh=None):
        """Every image of the batch: element i equals forward(x[i:i+1]) of a frozen model.  sizes_wh ([B,2] original
        (width, height)): boxes come back in pixels of the original images -- the evaluators'
        `bboxes *= scale` (utils/vocapi_evaluator_mask.py:71-72) done on the GPU for the whole batch."""
        if self.trainable:
            raise NotImplementedError("yolo355 is an inference engine: the training branch "
                                      "(models/slim_yolo_v2.py:360-382) is out of scope")
        if not quantization:
            # models/slim_yolo_v2.py:212-358 with the default kwarg: every tracker returns its input (:17-18), so the
            # forward is conv + bias + LeakyReLU(0.125) / max-pool in fp32 on the loaded (possibly dyadic) weights; `find`
            # divides by 2^r weights that were multiplied by 2^r (:222-227), which cancels up to fp32 rounding.
            # Runs as the SlimYOLOv2 graph without BatchNorm on the bf16 MFMA (fp32 accumulation): tolerance parity.
            net = self._get_f32_net(int(x.shape[0]), find)
            net.set_thresholds(self.conf_thresh, self.nms_thresh)
            dets = net.forward(x)
            if sizes_wh is not None:                        # the evaluators' `bboxes *= scale` (utils/vocapi_evaluator.py:69-70)
                wh = np.asarray(sizes_wh, np.float32).reshape(-1, 2)
                dets = [(b * np.array([[w, h, w, h]], np.float32), s, c) for (b, s, c), (w, h) in zip(dets, wh)]
            return dets
        B = int(x.shape[0])
        trackers = self._tracker_states()
        freeze = not self.trainable
        if any(t.first_a == 0 for t in trackers) or not freeze:
            # first call ever (:25-27) calibrates every tracker on this input, layer by layer, on ONE handle holding the whole batch
            eng = self._get_engine(B, find)
            sa = eng.calibrate(x, trackers, freeze=freeze)
            self._store_trackers(trackers)
        else:
            sa = [t.exponent() for t in trackers]
        if B > self.PIPELINE_CHUNK:
            # more than one batch-64 forward: chunks in flight on the pipeline's handles (y355_pipeline), results in order
            pipe = self._get_pipeline(find)
            pipe.set_act_exponents(sa)
            pipe.set_thresholds(self.conf_thresh, self.nms_thresh)
            return pipe.forward(x, find=find, sizes_wh=sizes_wh)
        eng = self._get_engine(B, find)
        eng.set_act_exponents(sa)
        eng.set_thresholds(self.conf_thresh, self.nms_thresh)
        if sizes_wh is not None:
            return eng.forward_scaled(x, sizes_wh, find=find)
        return eng.forward(x, find=find)

    def submit_batch(self, x, quantization=True, find=False, sizes_wh=None):
        """Asynchronous forward_batch for callers that have host work per batch (the evaluator loops,
        utils/vocapi_evaluator_mask.py:57-82): enqueue the batch on the pipeline and return a token at once; collect_batch(token)
        gives what forward_batch would have returned.  Up to `pipeline depth` (8) chunks of PIPELINE_CHUNK images may be
        outstanding; the trackers must be calibrated (one forward(x, quantization=True) before)."""
        if not quantization or self.trainable:
            raise NotImplementedError("submit_batch runs the int8 path (quantization=True, eval)")
        trackers = self._tracker_states()
        if any(t.first_a == 0 for t in trackers):
            raise RuntimeError("yolo355: calibrate the trackers with one forward(x, quantization=True) first")
        pipe = self._get_pipeline(find)
        pipe.set_act_exponents([t.exponent() for t in trackers])
        pipe.set_thresholds(self.conf_thresh, self.nms_thresh)
        xd = pipe._dev_input(x)
        n = int(xd.shape[0])
        if n > pipe.depth * pipe.max_batch:
            raise ValueError("submit_batch takes at most %d images at a time" % (pipe.depth * pipe.max_batch))
        wh = None
        if sizes_wh is not None:
            wh = torch.as_tensor(np.asarray(sizes_wh, np.float32).reshape(-1, 2)).to(pipe.device)
        tickets = []
        for i0 in range(0, n, pipe.max_batch):
            t = pipe.submit(xd[i0:i0 + pipe.max_batch], 0)
            if wh is not None:
                pipe.scale_boxes(t, wh[i0:i0 + pipe.max_batch])
            tickets.append(t)
        return (pipe, tickets, wh)

    def collect_batch(self, token):
        pipe, tickets, _ = token
        out = []
        for t in tickets:
            out.extend(pipe.fetch(t))
        return out

    def calibrate(self, x, freeze=False, find=False):
        """One calibration step on a batch, the tracker side of the reference's calibration loop
        (retune_bias_quantize.py:357-369: `model(images, target, quantization=True)` in training mode, whose loss branch is
        out of scope here): every AveragedRangeTracker sees max|activation| over the batch -- first call: scale =
        127 / max; later calls: scale = 0.9 scale + 0.1 * 127 / max (models/slim_yolo_v2.py:25-31), or unchanged with
        freeze=True -- layer by layer on the GPU, each layer running with the exponent just updated.  The buffers
        `a_tracker*.scale / first_a` are updated in place (they travel in the state_dict).  Returns the 11 exponents.
        The weights must already be power-of-two quantized (prep.quantize_layers): the reference's loop runs its first
        batch on the un-quantized weights and quantizes afterwards, which moves the first scales by < 2 % (measured,
        tests/test_round2.py) and none of the exponents of the goldens."""
        eng = self._get_engine(int(x.shape[0]), find)
        trackers = self._tracker_states()
        sa = eng.calibrate(x, trackers, freeze=freeze)
        self._store_trackers(trackers)
        return sa

    def forward_frames(self, frames, find=False):
        """New: detections for camera frames, uint8 [B,H,W,3] BGR at the network size --
        BaseTransform + BGR->RGB + HWC->CHW (data/__init__.py:30-56, test.py:79) run inside the first
        layer.  Element i equals forward(x_i) for the tensor the reference's transform makes of frame i.
        The trackers must be calibrated (one forward on a normalised tensor) beforehand."""
        eng = self._get_engine(int(frames.shape[0]), find)
        trackers = self._tracker_states()
        if any(t.first_a == 0 for t in trackers):
            raise RuntimeError("yolo355: calibrate the trackers with one forward(x, quantization=True) first")
        eng.set_act_exponents([t.exponent() for t in trackers])
        eng.set_thresholds(self.conf_thresh, self.nms_thresh)
        return eng.forward_frames(frames, find=find)

    # ------------------------------------------------------------------ engine plumbing
    def _weights_version(self):
        return tuple(int(p._version) for p in self.parameters()) + tuple(p.data_ptr() for p in self.parameters())

    def _get_engine(self, batch, find):
        key = (tuple(self.input_size), self.num_classes, tuple(map(tuple, self.anchor_size.tolist())))
        if self._engine is None or self._engine_key != key or self._engine.max_batch < batch:
            if self._engine is not None:
                self._engine.close()
            dev = self.device if isinstance(self.device, (str, torch.device)) else "cuda:0"
            self._engine = Engine(self.input_size, self.num_classes, self.anchor_size.tolist(),
                                  self.conf_thresh, self.nms_thresh, max_batch=max(batch, 1), device=dev)
            self._engine_key = key
            self._loaded_version = None
        ver = self._weights_version()
        if self._loaded_version != ver or self._loaded_find != find:
            self._load_into(self._engine, find)
            self._loaded_version, self._loaded_find = ver, find
        return self._engine

    def _load_into(self, target, find):
        """the ten dyadic layers of this model into an Engine or a Pipeline"""
        mods = [getattr(self, n).convs[0] for n in _CONVS] + [self.pred]
        for i, m in enumerate(mods):
            q_w, e_w = prep.as_dyadic_int8(m.weight)
            q_b, e_b = prep.as_dyadic_int8(m.bias)
            # find=True: the checkpoint holds W * 2^r and the reference divides the conv output
            # by 2^r (:222-227): same values as exponents e + r
            r = prep.RETUNE[i] if find else 0
            target.load_layer(i, q_w, q_b, e_w + r, e_b + r)
        target.set_retune(prep.RETUNE)

    def _get_pipeline(self, find):
        key = (tuple(self.input_size), self.num_classes, tuple(map(tuple, self.anchor_size.tolist())))
        if self._pipe is None or self._pipe_key != key:
            if self._pipe is not None:
                self._pipe.close()
            dev = self.device if isinstance(self.device, (str, torch.device)) else "cuda:0"
            self._pipe = Pipeline(self.input_size, self.num_classes, self.anchor_size.tolist(), self.conf_thresh, self.nms_thresh,
                                  max_batch=self.PIPELINE_CHUNK, device=dev)
            self._pipe_key, self._pipe_loaded = key, None
        ver = (self._weights_version(), bool(find))
        if self._pipe_loaded != ver:
            self._pipe.sync()
            self._load_into(self._pipe, find)
            self._pipe_loaded = ver
        return self._pipe

    def _get_f32_net(self, batch, find):
        """y355_net (Y355_ARCH_SLIM_V2, bf16) loaded with this model's conv weights and biases as they are."""
        key = (tuple(self.input_size), self.num_classes, tuple(map(tuple, self.anchor_size.tolist())))
        st = self.__dict__.setdefault("_f32", dict(net=None, key=None, ver=None))
        if st["net"] is None or st["key"] != key or st["net"].max_batch < batch:
            if st["net"] is not None:
                st["net"].close()
            dev = self.device if isinstance(self.device, (str, torch.device)) else "cuda:0"
            st["net"] = Net("slim_yolo_v2", self.input_size, self.num_classes, self.anchor_size.tolist(), self.conf_thresh,
                            self.nms_thresh, max_batch=max(batch, 1), device=dev, dtype="bf16")
            st["key"], st["ver"] = key, None
        ver = (self._weights_version(), bool(find))
        if st["ver"] != ver:
            mods = [getattr(self, n).convs[0] for n in _CONVS] + [self.pred]
            for i, m in enumerate(mods):
                w = m.weight.detach().float().cpu().numpy()
                b = m.bias.detach().float().cpu().numpy()
                if find:                                    # the checkpoint holds W * 2^r, the forward divides by 2^r
                    sc = np.float32(2.0 ** -prep.RETUNE[i])
                    w, b = w * sc, b * sc
                st["net"].load_layer(i, w, b)
            st["ver"] = ver
        return st["net"]

    def _tracker_states(self):
        return [prep.RangeTracker(getattr(self, n).scale, int(getattr(self, n).first_a.item())) for n in _TRACKERS]

    def _store_trackers(self, trackers):
        with torch.no_grad():
            for n, t in zip(_TRACKERS, trackers):
                m = getattr(self, n)
                m.scale.copy_(t.scale.to(m.scale.device))
                m.first_a.fill_(float(t.first_a))


class _NetModel(nn.Module):
    """Shared plumbing of the fp32 model drop-ins: fold BN, load the engine, batched forward."""
    _arch = None

    def _conv_modules(self):
        raise NotImplementedError

    def _flat_anchors(self):
        return [list(map(float, a)) for a in self.anchor_size.view(-1, 2).tolist()]

    def set_grid(self, input_size):
        self.input_size = list(input_size)
        self.scale = np.array([[[input_size[1], input_size[0], input_size[1], input_size[0]]]])

    def forward_batch(self, x, quantization=False):
        """Every image of the batch.  quantization=True (YOLOv3tiny only) runs the int8 engine:
        weights quantized per tensor to power-of-two int8 after the BN fold, activation exponents
        frozen at the first quantized call from the bf16 run of that input -- the first-call rule of
        AveragedRangeTracker (models/slim_yolo_v2.py:25-27) applied to this graph."""
        if self.trainable:
            raise NotImplementedError("yolo355 is an inference engine: the training branch is out of scope")
        if self.training and any(isinstance(m, nn.BatchNorm2d) for m in self.modules()):
            raise NotImplementedError("yolo355 folds BatchNorm with its running statistics: call .eval() first")
        net = self._get_net(int(x.shape[0]))
        if not quantization:
            net.set_thresholds(self.conf_thresh, self.nms_thresh)
            return net.forward(x)
        if self.act_exponents is None:
            self.act_exponents = net.calibration_exponents(x)
        qnet = self._get_net(int(x.shape[0]), int8=True)
        qnet.set_act_exponents(*self.act_exponents)
        qnet.set_thresholds(self.conf_thresh, self.nms_thresh)
        return qnet.forward(x)

    def _weights_version(self):
        t = list(self.parameters()) + list(self.buffers())
        return tuple(int(p._version) for p in t) + tuple(p.data_ptr() for p in t)

    act_exponents = None      # (sa_in, [sa per tensor]) of the int8 path, frozen at the first quantized call

    def _get_net(self, batch, int8=False):
        slot = "q" if int8 else "f"
        st = self.__dict__.setdefault("_nets", {}).setdefault(slot, dict(net=None, key=None, ver=None))
        key = (tuple(self.input_size), self.num_classes, tuple(map(tuple, self._flat_anchors())))
        if st["net"] is None or st["key"] != key or st["net"].max_batch < batch:
            if st["net"] is not None:
                st["net"].close()
            dev = self.device if isinstance(self.device, (str, torch.device)) else "cuda:0"
            st["net"] = Net(self._arch, self.input_size, self.num_classes, self._flat_anchors(), self.conf_thresh,
                            self.nms_thresh, max_batch=max(batch, 1), device=dev, dtype="int8" if int8 else "bf16")
            st["key"], st["ver"] = key, None
        ver = self._weights_version()
        if st["ver"] != ver:
            folded = [folded_f32(m) for m in self._conv_modules()]
            if int8:
                for i, q in enumerate(prep.quantize_folded(folded)):
                    st["net"].load_layer_i8(i, q["q_w"], q["q_b"], q["e_w"], q["e_b"])
            else:
                for i, (w, b) in enumerate(folded):
                    st["net"].load_layer(i, w, b)
            st["ver"] = ver
        return st["net"]


class SlimYOLOv2(_NetModel):
    """Drop-in for the reference's fp32 model (models/slim_yolo_v2.py:385-622): same constructor,
    modules (utils.modules.Conv2d = conv + BN + LeakyReLU(0.125)), state_dict keys and eval-mode
    return.  Inference runs BN-folded on the MI355X with bf16 MFMA and fp32 accumulation
    (include/yolo355.h, y355_net); `quantization` / `find` are accepted and ignored like the
    reference does (:549)."""
    _arch = "slim_yolo_v2"

    def __init__(self, device, input_size=None, num_classes=20, trainable=False, conf_thresh=0.01,
                 nms_thresh=0.5, anchor_size=None, hr=False):
        super().__init__()
        self.device = device
        self.input_size = list(input_size)
        self.num_classes = num_classes
        self.trainable = trainable
        self.conf_thresh = conf_thresh
        self.nms_thresh = nms_thresh
        self.anchor_size = torch.tensor(anchor_size)
        self.anchor_number = len(anchor_size)
        self.stride = 16
        self.scale = np.array([[[input_size[1], input_size[0], input_size[1], input_size[0]]]])
        self.conv1 = Conv2d(3, 16, 3, 1, leakyReLU=True)
        self.pool1 = nn.MaxPool2d(2, 2)
        self.conv2 = Conv2d(16, 32, 3, 1, leakyReLU=True)
        self.pool2 = nn.MaxPool2d(2, 2)
        self.conv3_1 = Conv2d(32, 64, 3, 1, leakyReLU=True)
        self.conv3_2 = Conv2d(64, 64, 3, 1, leakyReLU=True)
        self.pool3 = nn.MaxPool2d(2, 2)
        self.conv4_1 = Conv2d(64, 128, 3, 1, leakyReLU=True)
        self.conv4_2 = Conv2d(128, 128, 3, 1, leakyReLU=True)
        self.pool4 = nn.MaxPool2d(2, 2)
        self.conv5 = Conv2d(128, 256, 3, 1, leakyReLU=True)
        self.conv6 = Conv2d(256, 256, 3, 1, leakyReLU=True)
        self.conv7 = Conv2d(256, 256, 3, 1, leakyReLU=True)
        self.pred = nn.Conv2d(256, self.anchor_number * (1 + 4 + self.num_classes), 3, 1, padding=1)

    def _conv_modules(self):
        return [getattr(self, n).convs for n in _CONVS] + [self.pred]

    def forward(self, x, target=None, quantization=False, find=False):
        """Eval-mode return of the reference (:585-601): detections of image 0."""
        return self.forward_batch(x)[0]
