"""Python handle over the C ABI (include/yolo355.h).  One Engine = one GPU + one stream.

The engine is the batched, integer implementation of
SlimYOLOv2_quantize_bnfuse.forward(x, quantization=True) (models/slim_yolo_v2.py:212-358).
PyTorch is used only for device memory and the stream; all compute is in libyolo355.so.
"""
import atexit
import ctypes as C
import weakref

import numpy as np
import torch

from . import _ffi
from .prep import RETUNE, RangeTracker

NUM_LAYERS = 10
LAYER_NAMES = ["conv1", "conv2", "conv3_1", "conv3_2", "conv4_1", "conv4_2", "conv5", "conv6", "conv7", "pred"]


# Handles that own HIP streams / events must be destroyed while the HIP runtime is still alive: at interpreter shutdown the
# order of module teardown is arbitrary, and a __del__ that reaches hipStreamDestroy after the runtime's own atexit handlers
# have run crashes the process.  Every live Engine / Pipeline is closed from one atexit hook registered at import (i.e. after
# torch's and before its teardown, atexit being LIFO).
_live = weakref.WeakSet()


@atexit.register
def _close_all():
    for o in list(_live):
        try:
            o.close()
        except Exception:
            pass


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise RuntimeError("yolo355 needs an MI355X (HIP) device; no GPU is visible and there is no CPU fallback")
    dev = torch.device(device if device is not None else "cuda:0")
    if dev.type != "cuda":
        raise RuntimeError("yolo355 engines live on a GPU; got device %r" % (device,))
    return torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())


class Engine:
    def __init__(self, input_size, num_classes, anchors, conf_thresh=0.01, nms_thresh=0.5,
                 max_batch=1, max_det=0, device=None):
        self._h = None
        lib = _ffi.lib()
        self.device = _require_gpu(device)
        self.input_size = [int(input_size[0]), int(input_size[1])]
        self.num_classes = int(num_classes)
        self.anchors = [[float(a), float(b)] for a, b in anchors]
        self.max_batch = int(max_batch)
        cfg = _ffi.Config()
        cfg.device_id = self.device.index
        cfg.height, cfg.width = self.input_size
        cfg.num_classes = self.num_classes
        cfg.num_anchors = len(self.anchors)
        for i, (w, h) in enumerate(self.anchors):
            cfg.anchors[2 * i], cfg.anchors[2 * i + 1] = w, h
        cfg.conf_thresh, cfg.nms_thresh = float(conf_thresh), float(nms_thresh)
        cfg.max_batch, cfg.max_det = self.max_batch, int(max_det)
        with torch.cuda.device(self.device):
            self._stream = torch.cuda.current_stream(self.device)
            # launch on torch's current stream (handle 0 = the default stream) so that torch
            # copies / allocations and the engine's kernels are ordered without extra syncs
            cfg.stream = C.c_void_p(self._stream.cuda_stream)
            cfg.own_stream = 0
            h = C.c_void_p()
            _ffi.check(lib.y355_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._lib = lib
        self._owned = True
        _live.add(self)
        self.max_det = lib.y355_max_det(h)
        self.num_anchors_total = lib.y355_num_anchors_total(h)
        self.conf_thresh, self.nms_thresh = float(conf_thresh), float(nms_thresh)
        self._out = None

    @classmethod
    def _borrowed(cls, handle, like, stream):
        """Engine view of a handle somebody else owns (a pipeline's handle i): the taps / statistics / profiling calls of the
        engine ABI on it.  `like` supplies the configuration; the view launches on the handle's own stream."""
        e = cls.__new__(cls)
        e._h, e._lib, e._owned = handle, like._lib, False
        e.device, e.input_size, e.num_classes, e.anchors = like.device, like.input_size, like.num_classes, like.anchors
        e.max_batch, e.max_det = like.max_batch, like.max_det
        e.num_anchors_total = e._lib.y355_num_anchors_total(handle)
        e.conf_thresh, e.nms_thresh = like.conf_thresh, like.nms_thresh
        e._stream = stream
        e._out = None
        return e

    def close(self):
        if self._h is not None:
            if self._owned:
                self._lib.y355_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ stream ordering
    # Every kernel goes to the stream that was current when the engine was built (the C ABI takes the stream at
    # creation).  A caller that has since switched streams (`with torch.cuda.stream(s)`) is ordered explicitly: the
    # engine's stream waits for the caller's pending work before a launch, and the caller's stream waits for the
    # engine's before it reads results.  Same stream: no-ops.
    def _enter(self):
        cur = torch.cuda.current_stream(self.device)
        if cur != self._stream:
            self._stream.wait_stream(cur)
        return cur

    def _leave(self, cur):
        if cur != self._stream:
            cur.wait_stream(self._stream)

    # ------------------------------------------------------------------ weights / exponents
    def load_layer(self, idx, q_w, q_b, e_w, e_b):
        qw = np.ascontiguousarray(q_w, dtype=np.int8)
        qb = np.ascontiguousarray(q_b, dtype=np.int32)
        if np.abs(np.asarray(q_w)).max() > 127:
            raise ValueError("|q_w| > 127")
        _ffi.check(self._lib.y355_load_layer(self._h, idx, qw.ctypes.data, qb.ctypes.data,
                                             qw.shape[0], qw.shape[1], int(e_w), int(e_b)))

    def load_quantized(self, qlayers):
        """qlayers: 10 dicts with q_w [cout,cin,3,3], q_b [cout], e_w, e_b (conv1..conv7, pred)."""
        for i, L in enumerate(qlayers):
            self.load_layer(i, L["q_w"], L["q_b"], L["e_w"], L["e_b"])

    def set_act_exponents(self, sa):
        arr = (C.c_int32 * 11)(*[int(v) for v in sa])
        _ffi.check(self._lib.y355_set_act_exponents(self._h, arr))

    def get_act_exponents(self):
        arr = (C.c_int32 * 11)()
        _ffi.check(self._lib.y355_get_act_exponents(self._h, arr))
        return list(arr)

    def set_retune(self, retune=RETUNE):
        arr = (C.c_int32 * 10)(*[int(v) for v in retune])
        _ffi.check(self._lib.y355_set_retune(self._h, arr))

    def set_option(self, option, value):
        """Engine options of include/yolo355.h, e.g. set_option(_ffi.OPT_FUSE_FRONT, 0): one launch per layer."""
        _ffi.check(self._lib.y355_set_option(self._h, int(option), int(value)))

    def set_thresholds(self, conf_thresh, nms_thresh):
        self.conf_thresh, self.nms_thresh = float(conf_thresh), float(nms_thresh)
        _ffi.check(self._lib.y355_set_thresholds(self._h, self.conf_thresh, self.nms_thresh))

    # ------------------------------------------------------------------ inputs
    def _dev_input(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        if x.dim() != 4 or x.shape[1] != 3 or list(x.shape[2:]) != self.input_size:
            raise ValueError("expected [B,3,%d,%d], got %s" % (self.input_size[0], self.input_size[1], tuple(x.shape)))
        if x.shape[0] > self.max_batch:
            raise ValueError("batch %d > max_batch %d" % (x.shape[0], self.max_batch))
        return x.to(device=self.device, dtype=torch.float32).contiguous()

    # ------------------------------------------------------------------ calibration
    def calibrate(self, x, trackers, freeze=True):
        """One calibration step: the tracker semantics of models/slim_yolo_v2.py:16-38 layer by layer on the GPU, in ONE call
        of the C ABI (y355_calibrate: the state machine and its float32 arithmetic live in the library).
        trackers: 11 prep.RangeTracker (input, conv1..conv7, pred) -- the checkpoint buffers scale / first_a; they are handed
        to the handle, updated there (first call: scale = 127/max; frozen: unchanged; else EMA) and read back in place.
        Leaves the engine's feature maps and exponents as the reference's forward would.  Returns the exponents."""
        xd = self._dev_input(x)
        sa, mx = _calibrate_call(self._lib.y355_set_trackers, self._lib.y355_get_trackers, self._lib.y355_calibrate, self._h,
                                 xd, trackers, freeze, self._enter)
        self.calib_max = mx                               # max|.| seen by each tracker in this calibration
        return sa

    def layer_stats(self, idx):
        st = _ffi.LayerStats()
        _ffi.check(self._lib.y355_layer_stats_get(self._h, idx, C.byref(st)))
        return dict(absmax_t=st.absmax_t, frac_bits=st.frac_bits, saturated=st.saturated, guard=st.guard)

    def get_feature(self, idx, batch):
        """int8 [B,C,Ho,Wo] output of layer idx of the last run (parity tap)."""
        cout = [16, 32, 64, 64, 128, 128, 256, 256, 256, len(self.anchors) * (5 + self.num_classes)][idx]
        div = [2, 4, 4, 8, 8, 16, 16, 16, 16, 16][idx]
        out = np.empty((batch, cout, self.input_size[0] // div, self.input_size[1] // div), dtype=np.int8)
        _ffi.check(self._lib.y355_get_feature(self._h, idx, batch, out.ctypes.data))
        return out

    # ------------------------------------------------------------------ the hot path
    def _buffers(self, B):
        if self._out is None or self._out[0].shape[0] < B:
            md = self.max_det
            self._out = (torch.empty((self.max_batch, md, 4), dtype=torch.float32, device=self.device),
                         torch.empty((self.max_batch, md), dtype=torch.float32, device=self.device),
                         torch.empty((self.max_batch, md), dtype=torch.int32, device=self.device),
                         torch.zeros((self.max_batch,), dtype=torch.int32, device=self.device))
        return self._out

    def forward_device(self, xd, flags=0, out=None):
        """Asynchronous batched forward on device tensors.  xd: CUDA float32 [B,3,H,W] contiguous.
        Returns (boxes [max_batch,max_det,4], scores, cls int32, count int32) device tensors,
        rows >= B / entries >= count[b] undefined."""
        B = xd.shape[0]
        ob, os_, oc, on = out if out is not None else self._buffers(B)
        cur = self._enter()
        _ffi.check(self._lib.y355_forward(self._h, xd.data_ptr(), B, int(flags), ob.data_ptr(), os_.data_ptr(),
                                          oc.data_ptr(), on.data_ptr()))
        self._leave(cur)
        return ob, os_, oc, on

    def forward_frames_device(self, frames, flags=0, out=None):
        """Asynchronous batched forward on camera frames: CUDA uint8 [B,H,W,3] BGR at the network size
        (BaseTransform fused into the first layer, y355_forward_u8)."""
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3 or not frames.is_cuda:
            raise ValueError("expected a CUDA uint8 tensor [B,h,w,3]")
        frames = frames.contiguous()
        B = frames.shape[0]
        if B > self.max_batch:
            raise ValueError("batch %d > max_batch %d" % (B, self.max_batch))
        ob, os_, oc, on = out if out is not None else self._buffers(B)
        cur = self._enter()
        if list(frames.shape[1:3]) == self.input_size:
            _ffi.check(self._lib.y355_forward_u8(self._h, frames.data_ptr(), B, int(flags), ob.data_ptr(), os_.data_ptr(),
                                                 oc.data_ptr(), on.data_ptr()))
        else:       # frames of another size: BaseTransform's cv2.resize (data/__init__.py:36) on the GPU first
            _ffi.check(self._lib.y355_forward_u8_resized(self._h, frames.data_ptr(), int(frames.shape[1]), int(frames.shape[2]), B,
                                                         int(flags), ob.data_ptr(), os_.data_ptr(), oc.data_ptr(), on.data_ptr(), None))
        self._leave(cur)
        return ob, os_, oc, on

    def resize_frames(self, frames):
        """cv2.resize(image, (W, H)) of BaseTransform for a batch of uint8 [B,h,w,3] frames on the GPU (the stage
        y355_forward_u8_resized runs in front of the network); returns a CUDA uint8 tensor [B,H,W,3]."""
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(frames)
        fd = frames.to(self.device).contiguous()
        B = fd.shape[0]
        if B > self.max_batch:
            raise ValueError("batch %d > max_batch %d" % (B, self.max_batch))
        out = torch.empty((B, self.input_size[0], self.input_size[1], 3), dtype=torch.uint8, device=self.device)
        cur = self._enter()
        _ffi.check(self._lib.y355_forward_u8_resized(self._h, fd.data_ptr(), int(fd.shape[1]), int(fd.shape[2]), B, 0,
                                                     None, None, None, None, out.data_ptr()))
        self._leave(cur)
        return out

    def forward_frames(self, frames, find=False):
        """frames: uint8 [B,H,W,3] BGR (numpy or torch).  Same return as forward(normalised tensor)."""
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(frames)
        fd = frames.to(self.device)
        B = fd.shape[0]
        ob, os_, oc, on = self.forward_frames_device(fd, _ffi.F_GUARD if find else 0)
        n = on[:B].cpu().numpy()
        if find:
            sat, guard = self.counters()
            if guard:
                print("too high!!!")
                raise AssertionError("conv output exceeds the 16-bit head-room (find=True): %d positions" % guard)
        boxes, scores, cls = ob[:B].cpu().numpy(), os_[:B].cpu().numpy(), oc[:B].cpu().numpy()
        return [(boxes[i, :n[i]].copy(), scores[i, :n[i]].copy(), cls[i, :n[i]].astype(np.int64))
                for i in range(B)]

    def forward_scaled(self, x, sizes_wh, find=False, frames=False):
        """Batched forward + the evaluators' `bboxes *= [[w, h, w, h]]` on the GPU (y355_scale_boxes).
        x: float32 [B,3,H,W] (or uint8 frames [B,H,W,3] with frames=True); sizes_wh: [B,2] original (width, height).
        Returns a list over the batch of (bboxes in pixels of the original image, scores, cls_inds)."""
        wh = torch.as_tensor(np.asarray(sizes_wh, np.float32).reshape(-1, 2)).to(self.device)
        flags = _ffi.F_GUARD if find else 0
        if frames:
            fd = (torch.from_numpy(x) if isinstance(x, np.ndarray) else x).to(self.device)
            B = fd.shape[0]
            ob, os_, oc, on = self.forward_frames_device(fd, flags)
        else:
            xd = self._dev_input(x)
            B = xd.shape[0]
            ob, os_, oc, on = self.forward_device(xd, flags)
        if wh.shape[0] != B:
            raise ValueError("sizes_wh has %d rows for a batch of %d" % (wh.shape[0], B))
        cur = self._enter()
        _ffi.check(self._lib.y355_scale_boxes(self._h, ob.data_ptr(), on.data_ptr(), wh.data_ptr(), B))
        self._leave(cur)
        n = on[:B].cpu().numpy()
        if find:
            sat, guard = self.counters()
            if guard:
                print("too high!!!")
                raise AssertionError("conv output exceeds the 16-bit head-room (find=True): %d positions" % guard)
        boxes, scores, cls = ob[:B].cpu().numpy(), os_[:B].cpu().numpy(), oc[:B].cpu().numpy()
        return [(boxes[i, :n[i]].copy(), scores[i, :n[i]].copy(), cls[i, :n[i]].astype(np.int64)) for i in range(B)]

    def set_normalization(self, mean_bgr, std_bgr):
        m = (C.c_float * 3)(*[float(v) for v in mean_bgr])
        sd = (C.c_float * 3)(*[float(v) for v in std_bgr])
        _ffi.check(self._lib.y355_set_normalization(self._h, m, sd))

    def forward(self, x, find=False, tap=False):
        """list of (bboxes float32 [n,4], scores float32 [n], cls_inds int64 [n]) per image,
        in anchor-index order: the reference's eval-mode return for every image of the batch."""
        xd = self._dev_input(x)
        B = xd.shape[0]
        flags = (_ffi.F_GUARD if find else 0) | (_ffi.F_TAP if tap else 0)
        ob, os_, oc, on = self.forward_device(xd, flags)
        n = on[:B].cpu().numpy()
        if find:
            sat, guard = self.counters()
            if guard:
                print("too high!!!")
                raise AssertionError("conv output exceeds the 16-bit head-room (find=True): %d positions" % guard)
        boxes, scores, cls = ob[:B].cpu().numpy(), os_[:B].cpu().numpy(), oc[:B].cpu().numpy()
        return [(boxes[i, :n[i]].copy(), scores[i, :n[i]].copy(), cls[i, :n[i]].astype(np.int64))
                for i in range(B)]

    def counters(self):
        s, g = C.c_int64(), C.c_int64()
        _ffi.check(self._lib.y355_forward_counters(self._h, C.byref(s), C.byref(g)))
        return s.value, g.value

    def candidates(self, batch):
        N = self.num_anchors_total
        b = np.empty((batch, N, 4), np.float32)
        s = np.empty((batch, N), np.float32)
        c = np.empty((batch, N), np.int32)
        _ffi.check(self._lib.y355_get_candidates(self._h, batch, b.ctypes.data, s.ctypes.data, c.ctypes.data))
        return b, s, c

    def head_nms(self, pred_q, sa_pred):
        """Head only (slim_yolo_v2.py:330-358) on an int8 pred tensor [B,A*(5+C),Hs,Ws]."""
        pq = np.ascontiguousarray(pred_q, dtype=np.int8)
        B, md = pq.shape[0], self.max_det
        b = np.empty((B, md, 4), np.float32)
        s = np.empty((B, md), np.float32)
        c = np.empty((B, md), np.int32)
        n = np.empty((B,), np.int32)
        _ffi.check(self._lib.y355_head_nms(self._h, pq.ctypes.data, B, int(sa_pred), b.ctypes.data, s.ctypes.data,
                                           c.ctypes.data, n.ctypes.data))
        return [(b[i, :n[i]].copy(), s[i, :n[i]].copy(), c[i, :n[i]].astype(np.int64)) for i in range(B)]

    def sync(self):
        _ffi.check(self._lib.y355_sync(self._h))

    def profile(self, enable=True):
        """True / 1: intervals between HIP events around every launch (profile_ms); 2: also the launches' own start / end
        timestamps (profile_kernel_ms; the intervals of such a run are wider)."""
        _ffi.check(self._lib.y355_profile(self._h, int(enable)))

    def profile_ms(self):
        arr = (C.c_float * _ffi.NUM_TIMERS)()
        _ffi.check(self._lib.y355_profile_get(self._h, arr))
        return list(arr)

    def profile_kernel_ms(self):
        """the layers' own kernel durations of the last profiled forward (ms; 0 where the launch records none):
        start / end timestamps of the launch itself, i.e. what rocprofv3 reports, without the gaps between launches."""
        arr = (C.c_float * 10)()
        _ffi.check(self._lib.y355_profile_kernel_get(self._h, arr))
        return list(arr)

    def profile_kernels_ms(self):
        """own durations (ms) of EVERY launch of the last forward profiled with profile(2): slots 0..9 layers (the fused front
        end = slot 0), 10 head decode, 11 candidate sort, 12 NMS pair walk, 13 NMS rounds + output."""
        arr = (C.c_float * _ffi.NUM_KERNEL_TIMERS)()
        _ffi.check(self._lib.y355_profile_kernels_get(self._h, arr))
        return list(arr)


def _calibrate_call(set_fn, get_fn, cal_fn, handle, xd, trackers, freeze, enter):
    """push the 11 tracker states into the handle, run the C calibration step, read the states back into `trackers`"""
    if len(trackers) != 11:
        raise ValueError("expected 11 trackers (input, conv1..conv7, pred)")
    moms = {float(t.momentum) for t in trackers}
    if len(moms) != 1:
        raise ValueError("the 11 trackers must share one momentum")
    scale = (C.c_float * 11)(*[float(t.scale.reshape(-1)[0].item()) for t in trackers])
    first = (C.c_int32 * 11)(*[int(t.first_a) for t in trackers])
    _ffi.check(set_fn(handle, scale, first))
    sa = (C.c_int32 * 11)()
    mx = (C.c_float * 11)()
    enter()                                               # the call synchronises the engine's stream itself
    _ffi.check(cal_fn(handle, xd.data_ptr(), int(xd.shape[0]), 1 if freeze else 0, moms.pop(), sa, mx))
    _ffi.check(get_fn(handle, scale, first))
    for i, t in enumerate(trackers):
        t.scale = torch.tensor([scale[i]], dtype=torch.float32)
        t.first_a = int(first[i])
    return list(sa), [float(v) for v in mx]


class Pipeline:
    """The throughput regime behind the API (y355_pipeline, include/yolo355.h): `handles` engines on as many HIP streams, the
    submitted batches dealt to them round-robin, so that the head / NMS of one batch runs beside the convolutions of the next.

        pipe = Pipeline([416, 416], 2, ANCHOR_SIZE_MASK, max_batch=64)
        pipe.load_quantized(qlayers); pipe.calibrate(x0, trackers)
        t = pipe.submit(x_cuda)               # asynchronous; up to pipe.depth tickets in flight
        boxes, scores, cls, count = pipe.outputs(t)     # device tensors, valid after pipe.wait(t)
        dets = pipe.fetch(t)                  # or: the reference's per-image (bboxes, scores, cls_inds) lists on the host

    Results equal a stand-alone Engine's bit for bit (tests/test_pipeline.py).  forward(x) takes any batch size: chunks of
    max_batch in flight, results in order -- what models.SlimYOLOv2_quantize_bnfuse.forward_batch and the batched evaluators run."""

    def __init__(self, input_size, num_classes, anchors, conf_thresh=0.01, nms_thresh=0.5, max_batch=64, max_det=0,
                 device=None, handles=0, ring_workgroups=-1):
        self._h = None
        lib = _ffi.lib()
        self.device = _require_gpu(device)
        self.input_size = [int(input_size[0]), int(input_size[1])]
        self.num_classes = int(num_classes)
        self.anchors = [[float(a), float(b)] for a, b in anchors]
        self.max_batch = int(max_batch)
        cfg = _ffi.Config()
        cfg.device_id = self.device.index
        cfg.height, cfg.width = self.input_size
        cfg.num_classes = self.num_classes
        cfg.num_anchors = len(self.anchors)
        for i, (w, h) in enumerate(self.anchors):
            cfg.anchors[2 * i], cfg.anchors[2 * i + 1] = w, h
        cfg.conf_thresh, cfg.nms_thresh = float(conf_thresh), float(nms_thresh)
        cfg.max_batch, cfg.max_det = self.max_batch, int(max_det)
        h = C.c_void_p()
        nh = _ffi.PIPE_DEFAULT_HANDLES if int(handles) == 0 else int(handles)     # the C ABI's default (Y355_PIPE_DEFAULT_HANDLES)
        # the handles run on torch streams: PyTorch's allocator tracks memory per stream, so torch must own (and outlive)
        # every stream its tensors are used on -- y355_pipeline_create_on
        self._tstreams = [torch.cuda.Stream(device=self.device) for _ in range(max(nh, 1))]
        sp = (C.c_void_p * len(self._tstreams))(*[st.cuda_stream for st in self._tstreams])
        with torch.cuda.device(self.device):
            _ffi.check(lib.y355_pipeline_create_on(C.byref(cfg), nh, int(ring_workgroups), sp, C.byref(h)))
        self._h, self._lib = h, lib
        _live.add(self)
        self.handles = lib.y355_pipeline_handles(h)
        self.depth = lib.y355_pipeline_depth(h)
        self.max_det = lib.y355_pipeline_max_det(h)
        self.conf_thresh, self.nms_thresh = float(conf_thresh), float(nms_thresh)
        self._bufs = [None] * self.depth                  # torch-owned output buffers, one set per ticket slot
        self._tickets = {}                                # slot -> (ticket, batch, outputs, input) of the ticket that holds it
        self._next = 0

    def close(self):
        if self._h is not None:
            self._lib.y355_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ configuration (every handle)
    def load_layer(self, idx, q_w, q_b, e_w, e_b):
        qw = np.ascontiguousarray(q_w, dtype=np.int8)
        qb = np.ascontiguousarray(q_b, dtype=np.int32)
        if np.abs(np.asarray(q_w)).max() > 127:
            raise ValueError("|q_w| > 127")
        _ffi.check(self._lib.y355_pipeline_load_layer(self._h, idx, qw.ctypes.data, qb.ctypes.data, qw.shape[0], qw.shape[1],
                                                      int(e_w), int(e_b)))

    def load_quantized(self, qlayers):
        for i, L in enumerate(qlayers):
            self.load_layer(i, L["q_w"], L["q_b"], L["e_w"], L["e_b"])

    def set_act_exponents(self, sa):
        _ffi.check(self._lib.y355_pipeline_set_act_exponents(self._h, (C.c_int32 * 11)(*[int(v) for v in sa])))

    def set_retune(self, retune=RETUNE):
        _ffi.check(self._lib.y355_pipeline_set_retune(self._h, (C.c_int32 * 10)(*[int(v) for v in retune])))

    def set_thresholds(self, conf_thresh, nms_thresh):
        self.conf_thresh, self.nms_thresh = float(conf_thresh), float(nms_thresh)
        _ffi.check(self._lib.y355_pipeline_set_thresholds(self._h, self.conf_thresh, self.nms_thresh))

    def set_option(self, option, value):
        _ffi.check(self._lib.y355_pipeline_set_option(self._h, int(option), int(value)))

    def set_normalization(self, mean_bgr, std_bgr):
        _ffi.check(self._lib.y355_pipeline_set_normalization(self._h, (C.c_float * 3)(*[float(v) for v in mean_bgr]),
                                                             (C.c_float * 3)(*[float(v) for v in std_bgr])))

    def calibrate(self, x, trackers, freeze=True):
        """Engine.calibrate on handle 0 (y355_pipeline_calibrate); every handle gets the resulting trackers and exponents."""
        xd = self._dev_input(x)
        sa, mx = _calibrate_call(self._lib.y355_pipeline_set_trackers, self._lib.y355_pipeline_get_trackers,
                                 self._lib.y355_pipeline_calibrate, self._h, xd, trackers, freeze,
                                 lambda: torch.cuda.current_stream(self.device).synchronize())
        self.calib_max = mx
        return sa

    # ------------------------------------------------------------------ the hot path
    def _dev_input(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        if x.dim() != 4 or x.shape[1] != 3 or list(x.shape[2:]) != self.input_size:
            raise ValueError("expected [B,3,%d,%d], got %s" % (self.input_size[0], self.input_size[1], tuple(x.shape)))
        return x.to(device=self.device, dtype=torch.float32).contiguous()

    def _slot_bufs(self, slot):
        if self._bufs[slot] is None:
            md, B = self.max_det, self.max_batch
            self._bufs[slot] = (torch.empty((B, md, 4), dtype=torch.float32, device=self.device),
                                torch.empty((B, md), dtype=torch.float32, device=self.device),
                                torch.empty((B, md), dtype=torch.int32, device=self.device),
                                torch.zeros((B,), dtype=torch.int32, device=self.device))
        return self._bufs[slot]

    def submit(self, xd, flags=0, out=None, frames=False, ordered=True):
        """Enqueue one forward; returns the ticket.  xd: CUDA float32 [B,3,H,W] contiguous (frames=True: uint8 [B,H,W,3] BGR at
        the network size).  out: (boxes, scores, cls int32, count int32) device tensors (default: the pipeline's buffer set of
        the ticket's slot, reused `depth` submits later).  ordered=True: the forward starts behind whatever torch's current
        stream has queued (the input's producer); False: the caller guarantees the input is complete."""
        B = int(xd.shape[0])
        if B > self.max_batch:
            raise ValueError("batch %d > max_batch %d" % (B, self.max_batch))
        slot = self._next % self.depth
        ob, os_, oc, on = out if out is not None else self._slot_bufs(slot)
        t = C.c_longlong()
        cur = torch.cuda.current_stream(self.device).cuda_stream if ordered else 0
        fn = self._lib.y355_pipeline_submit_u8 if frames else self._lib.y355_pipeline_submit
        _ffi.check(fn(self._h, xd.data_ptr(), B, int(flags) | (_ffi.PIPE_AFTER_STREAM if ordered else 0), cur,
                      ob.data_ptr(), os_.data_ptr(), oc.data_ptr(), on.data_ptr(), C.byref(t)))
        self._next = t.value + 1
        self._tickets[t.value % self.depth] = (t.value, B, (ob, os_, oc, on), xd)   # keeps the input alive until its slot is reused
        return t.value

    def outputs(self, ticket):
        """(boxes [max_batch,max_det,4], scores, cls, count) device tensors of the ticket; rows >= B / entries >= count[b] undefined.
        Read them after wait(ticket)."""
        rec = self._tickets.get(ticket % self.depth)
        if rec is None or rec[0] != ticket:
            raise _ffi.Y355Error(_ffi.ENOTREADY, "ticket %d is gone (a ticket lives for %d more submits)" % (ticket, self.depth))
        return rec[2]

    def wait(self, ticket, host=False):
        """host=False: torch's current stream waits for the ticket (no host block); True: the host blocks until it is done."""
        cur = torch.cuda.current_stream(self.device).cuda_stream
        _ffi.check(self._lib.y355_pipeline_wait(self._h, int(ticket), 0 if host else 1, cur))

    def release(self, ticket):
        """the work queued on torch's current stream so far is the last reader of the ticket's outputs"""
        _ffi.check(self._lib.y355_pipeline_release(self._h, int(ticket), torch.cuda.current_stream(self.device).cuda_stream))

    def stream(self, ticket):
        """torch view of the HIP stream of the handle that runs `ticket` (for work that must follow it without an event)"""
        return self._tstreams[int(ticket) % self.handles]

    def engine(self, i):
        """Engine view of handle i (not owning): taps, statistics, profiling, options.  Do not run forwards on it while tickets
        are in flight."""
        if not 0 <= i < self.handles:
            raise IndexError(i)
        return Engine._borrowed(C.c_void_p(self._lib.y355_pipeline_engine(self._h, i)), self, self._tstreams[i])

    @property
    def next_ticket(self):
        return self._next

    def scale_boxes(self, ticket, wh_dev):
        _ffi.check(self._lib.y355_pipeline_scale_boxes(self._h, int(ticket), wh_dev.data_ptr()))

    def fetch(self, ticket):
        """The reference's eval-mode return for every image of the ticket's batch: list of (bboxes float32 [n,4], scores
        float32 [n], cls_inds int64 [n]), anchor-index order (models/slim_yolo_v2.py:205-210)."""
        rec = self._tickets.get(ticket % self.depth)
        if rec is None or rec[0] != ticket:
            raise _ffi.Y355Error(_ffi.ENOTREADY, "ticket %d is gone (a ticket lives for %d more submits)" % (ticket, self.depth))
        B, md = rec[1], self.max_det
        b = np.empty((B, md, 4), np.float32)
        s = np.empty((B, md), np.float32)
        c = np.empty((B, md), np.int32)
        n = np.empty((B,), np.int32)
        _ffi.check(self._lib.y355_pipeline_fetch(self._h, int(ticket), b.ctypes.data, s.ctypes.data, c.ctypes.data, n.ctypes.data))
        return [(b[i, :n[i]].copy(), s[i, :n[i]].copy(), c[i, :n[i]].astype(np.int64)) for i in range(B)]

    def counters(self):
        s, g = C.c_int64(), C.c_int64()
        _ffi.check(self._lib.y355_pipeline_counters(self._h, C.byref(s), C.byref(g)))
        return s.value, g.value

    def forward(self, x, find=False, sizes_wh=None, frames=False):
        """Any number of images: chunks of max_batch submitted back to back (up to `depth` in flight), results in order.
        sizes_wh [B,2]: the evaluators' `bboxes *= [[w, h, w, h]]` on the GPU.  find=True: the 2^15 head-room guard."""
        if frames:
            xd = (torch.from_numpy(x) if isinstance(x, np.ndarray) else x).to(self.device).contiguous()
        else:
            xd = self._dev_input(x)
        n = int(xd.shape[0])
        wh = None
        if sizes_wh is not None:
            wh = torch.as_tensor(np.asarray(sizes_wh, np.float32).reshape(-1, 2)).to(self.device)
            if wh.shape[0] != n:
                raise ValueError("sizes_wh has %d rows for a batch of %d" % (wh.shape[0], n))
        flags = _ffi.F_GUARD if find else 0
        res, tickets = [], []
        guard = 0
        for i0 in range(0, n, self.max_batch):
            if len(tickets) == self.depth:                # the oldest ticket's slot is about to be reused: take its result first
                res.extend(self.fetch(tickets.pop(0)))
            t = self.submit(xd[i0:i0 + self.max_batch], flags, frames=frames)
            if wh is not None:
                self.scale_boxes(t, wh[i0:i0 + self.max_batch])
            tickets.append(t)
            if find:                                      # the guard count of THIS forward: its handle's counters, read before the next
                sat, g = C.c_int64(), C.c_int64()
                eh = C.c_void_p(self._lib.y355_pipeline_engine(self._h, t % self.handles))
                _ffi.check(self._lib.y355_forward_counters(eh, C.byref(sat), C.byref(g)))
                guard += g.value
        for t in tickets:
            res.extend(self.fetch(t))
        if find and guard:
            print("too high!!!")
            raise AssertionError("conv output exceeds the 16-bit head-room (find=True): %d positions" % guard)
        return res

    def sync(self):
        _ffi.check(self._lib.y355_pipeline_sync(self._h))


def mfma_peak_i8(device_id=0, ms_target=50.0):
    """(Top/s, in-kernel clock in GHz) of back-to-back v_mfma_i32_16x16x64_i8 on this GPU (y355_mfma_peak_i8)."""
    t, c = C.c_float(), C.c_float()
    _ffi.check(_ffi.lib().y355_mfma_peak_i8(int(device_id), float(ms_target), C.byref(t), C.byref(c)))
    return float(t.value), float(c.value)


class ConvOp:
    """A convolution whose weights live packed on the GPU (y355_conv_op, include/yolo355.h): the device-resident form of
    conv2d_bf16 / conv3x3_i8_raw for callers whose tensors are CUDA tensors -- forward takes and returns CUDA tensors on
    torch's current stream, no tensor crosses to the host."""

    def __init__(self, handle, kind, cout, stride, device):
        self._h, self.kind, self.cout, self.stride, self.device = handle, kind, int(cout), int(stride), device
        self._lib = _ffi.lib()
        _live.add(self)

    @classmethod
    def bf16(cls, w, bias, stride=1, neg_slope=1.0, device=None):
        dev = _require_gpu(device)
        wi = np.ascontiguousarray(w, dtype=np.float32)
        cout, cin, k, k2 = wi.shape
        if k != k2:
            raise ValueError("square kernels only")
        bi = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
        h = C.c_void_p()
        _ffi.check(_ffi.lib().y355_conv_op_create_bf16(dev.index, wi.ctypes.data, None if bi is None else bi.ctypes.data, cin, cout, k,
                                                       int(stride), float(neg_slope), C.byref(h)))
        return cls(h, "bf16", cout, stride, dev)

    @classmethod
    def int8(cls, q_w, q_b, e_w, e_b, leaky=True, relu=False, device=None):
        dev = _require_gpu(device)
        qw = np.ascontiguousarray(q_w, dtype=np.int8)
        qb = np.ascontiguousarray(q_b, dtype=np.int32)
        h = C.c_void_p()
        _ffi.check(_ffi.lib().y355_conv_op_create_i8(dev.index, qw.ctypes.data, qb.ctypes.data, qw.shape[1], qw.shape[0], int(e_w), int(e_b),
                                                     _act_flag(leaky, relu), C.byref(h)))
        return cls(h, "int8", qw.shape[0], 1, dev)

    def close(self):
        if self._h is not None:
            self._lib.y355_conv_op_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, x, residual=None, out_fp32=False):
        """bf16 form: x CUDA float32 [B,cin,H,W] -> CUDA float32 [B,cout,Ho,Wo] (asynchronous, torch's current stream)."""
        x = x.detach().to(dtype=torch.float32).contiguous()
        B, _, H, W = x.shape
        Ho, Wo = ((H + 1) // 2, (W + 1) // 2) if self.stride == 2 else (H, W)
        out = torch.empty((B, self.cout, Ho, Wo), dtype=torch.float32, device=x.device)
        r = None if residual is None else residual.detach().to(dtype=torch.float32).contiguous()
        if r is not None and tuple(r.shape) != tuple(out.shape):
            raise ValueError("residual shape %s, expected %s" % (tuple(r.shape), tuple(out.shape)))
        _ffi.check(self._lib.y355_conv_op_forward(self._h, x.data_ptr(), None if r is None else r.data_ptr(), B, H, W, 1 if out_fp32 else 0,
                                                  out.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream))
        return out

    def forward_i8(self, x):
        """int8 form: Conv2d_fuse on a dyadic CUDA tensor, exact; returns None when x is not a dyadic int8 tensor."""
        x = x.detach().to(dtype=torch.float32).contiguous()
        B, _, H, W = x.shape
        out = torch.empty((B, self.cout, H, W), dtype=torch.float32, device=x.device)
        sa, exact = C.c_int32(), C.c_int32()
        _ffi.check(self._lib.y355_conv_op_forward_i8(self._h, x.data_ptr(), B, H, W, out.data_ptr(),
                                                     torch.cuda.current_stream(x.device).cuda_stream, C.byref(sa), C.byref(exact)))
        return out if exact.value else None


def _dev_unary(fn_name, x, out_shape, *args):
    """one of the y355_*_f32_dev operators on a CUDA tensor, on torch's current stream"""
    x = x.detach().to(dtype=torch.float32).contiguous()
    out = torch.empty(out_shape, dtype=torch.float32, device=x.device)
    B, Cc, H, W = x.shape
    _ffi.check(getattr(_ffi.lib(), fn_name)(x.data_ptr(), B, Cc, H, W, *args, out.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream))
    return out


def reorg_f32_dev(x, stride):
    B, Cc, H, W = x.shape
    s = int(stride)
    return _dev_unary("y355_reorg_f32_dev", x, (B, Cc * s * s, H // s, W // s), s)


def spp_f32_dev(x):
    B, Cc, H, W = x.shape
    return _dev_unary("y355_spp_f32_dev", x, (B, 4 * Cc, H, W))


def maxpool2x2_f32_dev(x):
    B, Cc, H, W = x.shape
    return _dev_unary("y355_maxpool2x2_f32_dev", x, (B, Cc, H // 2, W // 2))


def upsample2x_f32_dev(x):
    B, Cc, H, W = x.shape
    return _dev_unary("y355_upsample2x_f32_dev", x, (B, Cc, 2 * H, 2 * W))


def _act_flag(leaky, relu):
    if leaky and relu:
        raise ValueError("LeakyReLU and ReLU are exclusive")
    return _ffi.OP_LEAKY if leaky else (_ffi.OP_RELU if relu else 0)


def conv3x3_i8_fused(q_in, q_w, q_b, sa_in, e_w, e_b, sa_out, leaky=True, pool=False, device_id=0, relu=False):
    """Operator-level fused layer (y355_conv3x3_i8_fused): numpy int8 NCHW in/out.  relu=True (with leaky=False):
    the ReLU epilogue of Conv2d_fuse(leakyReLU=False) (utils/modules.py:26)."""
    lib = _ffi.lib()
    if not torch.cuda.is_available():
        raise RuntimeError("yolo355 needs a GPU; there is no CPU fallback")
    qi = np.ascontiguousarray(q_in, dtype=np.int8)
    qw = np.ascontiguousarray(q_w, dtype=np.int8)
    qb = np.ascontiguousarray(q_b, dtype=np.int32)
    B, cin, H, W = qi.shape
    cout = qw.shape[0]
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = np.empty((B, cout, Ho, Wo), np.int8)
    st = _ffi.LayerStats()
    flags = _act_flag(leaky, relu) | (_ffi.OP_POOL if pool else 0)
    _ffi.check(lib.y355_conv3x3_i8_fused(int(device_id), qi.ctypes.data, qw.ctypes.data, qb.ctypes.data, B, cin, cout,
                                         H, W, int(sa_in), int(e_w), int(e_b), int(sa_out), flags,
                                         out.ctypes.data, C.byref(st)))
    return out, dict(absmax_t=st.absmax_t, frac_bits=st.frac_bits, saturated=st.saturated, guard=st.guard)


def conv3x3_i8_raw(q_in, q_w, q_b, sa_in, e_w, e_b, leaky=True, device_id=0, relu=False):
    """conv + bias + LeakyReLU(0.125) (or ReLU, or nothing) without requantisation (y355_conv3x3_i8_raw):
    returns (t' int64 [B,cout,H,W], F') with value = t' / 2^F'."""
    lib = _ffi.lib()
    if not torch.cuda.is_available():
        raise RuntimeError("yolo355 needs a GPU; there is no CPU fallback")
    qi = np.ascontiguousarray(q_in, dtype=np.int8)
    qw = np.ascontiguousarray(q_w, dtype=np.int8)
    qb = np.ascontiguousarray(q_b, dtype=np.int32)
    B, cin, H, W = qi.shape
    cout = qw.shape[0]
    out = np.empty((B, cout, H, W), np.int64)
    fb = C.c_int32()
    _ffi.check(lib.y355_conv3x3_i8_raw(int(device_id), qi.ctypes.data, qw.ctypes.data, qb.ctypes.data, B, cin, cout,
                                       H, W, int(sa_in), int(e_w), int(e_b), _act_flag(leaky, relu),
                                       out.ctypes.data, C.byref(fb)))
    return out, fb.value


def quantize_input_f32_i8(x, sa, device_id=0):
    """Stand-alone input fake-quant (y355_quantize_input_f32_i8): (q int8 like x, clamped count)."""
    lib = _ffi.lib()
    if not torch.cuda.is_available():
        raise RuntimeError("yolo355 needs a GPU; there is no CPU fallback")
    xf = np.ascontiguousarray(x, dtype=np.float32)
    q = np.empty(xf.shape, np.int8)
    c = C.c_int64()
    _ffi.check(lib.y355_quantize_input_f32_i8(int(device_id), xf.ctypes.data, xf.size, int(sa), q.ctypes.data, C.byref(c)))
    return q, c.value


def maxpool2x2_i8(q, device_id=0):
    """Stand-alone 2x2 / stride 2 max-pool on int8 NCHW (y355_maxpool2x2_i8)."""
    lib = _ffi.lib()
    if not torch.cuda.is_available():
        raise RuntimeError("yolo355 needs a GPU; there is no CPU fallback")
    qi = np.ascontiguousarray(q, dtype=np.int8)
    B, Cc, H, W = qi.shape
    out = np.empty((B, Cc, H // 2, W // 2), np.int8)
    _ffi.check(lib.y355_maxpool2x2_i8(int(device_id), qi.ctypes.data, B, Cc, H, W, out.ctypes.data))
    return out


def _need_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("yolo355 needs a GPU; there is no CPU fallback")


def reorg_f32(x, stride, device_id=0):
    """utils.modules.reorg_layer.forward on fp32 NCHW (y355_reorg_f32); bit-exact data movement."""
    lib = _ffi.lib()
    _need_gpu()
    xi = np.ascontiguousarray(x, dtype=np.float32)
    B, Cc, H, W = xi.shape
    s = int(stride)
    out = np.empty((B, Cc * s * s, H // max(s, 1), W // max(s, 1)), np.float32)
    _ffi.check(lib.y355_reorg_f32(int(device_id), xi.ctypes.data, B, Cc, H, W, s, out.ctypes.data))
    return out


def spp_f32(x, device_id=0):
    """utils.modules.SPP.forward on fp32 NCHW (y355_spp_f32); bit-exact."""
    lib = _ffi.lib()
    _need_gpu()
    xi = np.ascontiguousarray(x, dtype=np.float32)
    B, Cc, H, W = xi.shape
    out = np.empty((B, 4 * Cc, H, W), np.float32)
    _ffi.check(lib.y355_spp_f32(int(device_id), xi.ctypes.data, B, Cc, H, W, out.ctypes.data))
    return out


def conv2d_bf16(x, w, bias=None, residual=None, stride=1, neg_slope=1.0, out_fp32=False, device_id=0):
    """conv (1x1, or 3x3 pad 1; stride 1, or 2 for 3x3) + bias + LeakyReLU(neg_slope) [+ residual] on the bf16 MFMA
    (y355_conv2d_bf16).  fp32 NCHW in and out; operands and result are rounded to bf16."""
    lib = _ffi.lib()
    _need_gpu()
    xi = np.ascontiguousarray(x, dtype=np.float32)
    wi = np.ascontiguousarray(w, dtype=np.float32)
    B, Cin, H, W = xi.shape
    Cout, Cin2, k, k2 = wi.shape
    if Cin2 != Cin or k != k2:
        raise ValueError("weight shape %s does not match the input's %d channels" % (wi.shape, Cin))
    s = int(stride)
    Ho, Wo = ((H + 1) // 2, (W + 1) // 2) if s == 2 else (H, W)
    bi = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    ri = None if residual is None else np.ascontiguousarray(residual, dtype=np.float32)
    if ri is not None and ri.shape != (B, Cout, Ho, Wo):
        raise ValueError("residual shape %s, expected %s" % (ri.shape, (B, Cout, Ho, Wo)))
    out = np.empty((B, Cout, Ho, Wo), np.float32)
    _ffi.check(lib.y355_conv2d_bf16(int(device_id), xi.ctypes.data, wi.ctypes.data, None if bi is None else bi.ctypes.data,
                                    None if ri is None else ri.ctypes.data, B, Cin, Cout, H, W, int(k), s, float(neg_slope),
                                    1 if out_fp32 else 0, out.ctypes.data))
    return out


def maxpool2x2_f32(x, device_id=0):
    """nn.MaxPool2d((2, 2), 2) on fp32 NCHW (y355_maxpool2x2_f32); bit-exact."""
    lib = _ffi.lib()
    _need_gpu()
    xi = np.ascontiguousarray(x, dtype=np.float32)
    B, Cc, H, W = xi.shape
    out = np.empty((B, Cc, H // 2, W // 2), np.float32)
    _ffi.check(lib.y355_maxpool2x2_f32(int(device_id), xi.ctypes.data, B, Cc, H, W, out.ctypes.data))
    return out


def head_f32(preds, strides, anchors, num_classes, input_size, wh_mul, conf_thresh, nms_thresh, max_det=None, device_id=0):
    """Detection head on fp32 prediction maps (y355_head_f32).  preds: list (1 or 2 levels) of [B, A*(5+C), Hs, Ws];
    anchors: [nlev][A][2].  Returns a list over the batch of (boxes [n,4] normalised, scores [n], classes int64 [n])."""
    import ctypes as C
    lib = _ffi.lib()
    _need_gpu()
    ps = [np.ascontiguousarray(p, dtype=np.float32) for p in preds]
    nlev = len(ps)
    B = ps[0].shape[0]
    an = np.ascontiguousarray(anchors, dtype=np.float32).reshape(nlev, -1, 2)
    A = an.shape[1]
    hs = (C.c_int * nlev)(*[p.shape[2] for p in ps])
    ws = (C.c_int * nlev)(*[p.shape[3] for p in ps])
    st = (C.c_float * nlev)(*[float(s) for s in strides])
    ptrs = (C.c_void_p * nlev)(*[p.ctypes.data for p in ps])
    N = min(sum(p.shape[2] * p.shape[3] * A for p in ps), 4096)      # the head keeps at most 4096 candidates per image
    md = N if max_det is None else min(int(max_det), N)
    boxes = np.zeros((B, md, 4), np.float32)
    scores = np.zeros((B, md), np.float32)
    cls = np.zeros((B, md), np.int32)
    count = np.zeros((B,), np.int32)
    _ffi.check(lib.y355_head_f32(int(device_id), nlev, ptrs, hs, ws, st, an.ctypes.data_as(C.POINTER(C.c_float)), A, int(num_classes),
                                 int(input_size[0]), int(input_size[1]), float(wh_mul), float(conf_thresh), float(nms_thresh), B, md,
                                 boxes.ctypes.data, scores.ctypes.data, cls.ctypes.data, count.ctypes.data))
    return [(boxes[b, :count[b]].copy(), scores[b, :count[b]].copy(), cls[b, :count[b]].astype(np.int64)) for b in range(B)]


def upsample2x_f32(x, device_id=0):
    """F.interpolate(x, scale_factor=2.0, mode="bilinear", align_corners=True) on fp32 NCHW (y355_upsample2x_f32)."""
    lib = _ffi.lib()
    _need_gpu()
    xi = np.ascontiguousarray(x, dtype=np.float32)
    B, Cc, H, W = xi.shape
    out = np.empty((B, Cc, 2 * H, 2 * W), np.float32)
    _ffi.check(lib.y355_upsample2x_f32(int(device_id), xi.ctypes.data, B, Cc, H, W, out.ctypes.data))
    return out
