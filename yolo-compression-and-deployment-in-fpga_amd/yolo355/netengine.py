"""Python handle over the y355_net_* C ABI (include/yolo355.h): the table-driven executor of
csrc/net.hip for SlimYOLOv2 (fp32 model -> bf16 MFMA) and YOLOv3tiny.  PyTorch is used for
device memory and the stream only; all compute is in libyolo355.so."""
import ctypes as C

import numpy as np
import torch

from . import _ffi
from .engine import _require_gpu

ARCH = {"slim_yolo_v2": _ffi.ARCH_SLIM_V2, "tiny_yolo_v3": _ffi.ARCH_TINY_V3, "yolo_v2": _ffi.ARCH_YOLO_V2,
        "yolo_v3": _ffi.ARCH_YOLO_V3, "yolo_v3_spp": _ffi.ARCH_YOLO_V3_SPP}
NLEV = {"slim_yolo_v2": 1, "tiny_yolo_v3": 2, "yolo_v2": 1, "yolo_v3": 3, "yolo_v3_spp": 3}
DTYPE = {"int8": _ffi.DT_INT8, "bf16": _ffi.DT_BF16}


class Net:
    def __init__(self, arch, input_size, num_classes, anchors, conf_thresh=0.01, nms_thresh=0.5,
                 max_batch=1, max_det=0, device=None, dtype="bf16"):
        """anchors: [[w, h], ...] -- A pairs for slim_yolo_v2 (grid units), 2*A pairs for tiny_yolo_v3
        (pixels; the stride-16 level first, data/config.py:27-31)."""
        self._h = None
        lib = _ffi.lib()
        self.device = _require_gpu(device)
        self.arch = arch
        self.input_size = [int(input_size[0]), int(input_size[1])]
        self.num_classes = int(num_classes)
        self.anchors = [[float(a), float(b)] for a, b in anchors]
        nlev = NLEV[arch]
        if len(self.anchors) % nlev:
            raise ValueError("%s needs a multiple of %d anchors" % (arch, nlev))
        self.max_batch = int(max_batch)
        cfg = _ffi.NetConfig()
        cfg.device_id = self.device.index
        cfg.arch, cfg.dtype = ARCH[arch], DTYPE[dtype]
        cfg.height, cfg.width = self.input_size
        cfg.num_classes = self.num_classes
        cfg.num_anchors = len(self.anchors) // nlev
        if len(self.anchors) > _ffi.MAX_ANCHORS:
            raise ValueError("too many anchors")
        for i, (w, h) in enumerate(self.anchors):
            cfg.anchors[2 * i], cfg.anchors[2 * i + 1] = w, h
        cfg.conf_thresh, cfg.nms_thresh = float(conf_thresh), float(nms_thresh)
        cfg.max_batch, cfg.max_det = self.max_batch, int(max_det)
        with torch.cuda.device(self.device):
            self._stream = torch.cuda.current_stream(self.device)
            cfg.stream = C.c_void_p(self._stream.cuda_stream)
            cfg.own_stream = 0
            h = C.c_void_p()
            _ffi.check(lib.y355_net_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._lib = lib
        self.max_det = lib.y355_net_max_det(h)
        self.num_anchors_total = lib.y355_net_num_anchors_total(h)
        self.num_layers = lib.y355_net_num_layers(h)
        self.num_tensors = lib.y355_net_num_tensors(h)
        self._out = None

    def close(self):
        if self._h is not None:
            self._lib.y355_net_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def layer_shape(self, idx):
        s = (C.c_int32 * 4)()
        _ffi.check(self._lib.y355_net_layer_shape(self._h, idx, s))
        return tuple(s)

    def tensor_shape(self, idx):
        s = (C.c_int32 * 3)()
        _ffi.check(self._lib.y355_net_tensor_shape(self._h, idx, s))
        return tuple(s)

    def load_layer(self, idx, w, b=None):
        """w fp32 [cout,cin,k,k] (BN already folded), b fp32 [cout] or None."""
        w = np.ascontiguousarray(w, dtype=np.float32)
        bb = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
        _ffi.check(self._lib.y355_net_load_layer_f32(self._h, idx, w.ctypes.data, None if bb is None else bb.ctypes.data,
                                                     w.shape[0], w.shape[1], w.shape[2]))

    def load_layer_i8(self, idx, q_w, q_b, e_w, e_b):
        """int8 nets: q_w [cout,cin,k,k] (|q| <= 127, value q / 2^e_w), q_b int32 [cout] (value q / 2^e_b)."""
        if np.abs(np.asarray(q_w)).max() > 127:
            raise ValueError("|q_w| > 127")
        qw = np.ascontiguousarray(q_w, dtype=np.int8)
        qb = np.ascontiguousarray(q_b, dtype=np.int32)
        _ffi.check(self._lib.y355_net_load_layer_i8(self._h, idx, qw.ctypes.data, qb.ctypes.data, qw.shape[0], qw.shape[1],
                                                    qw.shape[2], int(e_w), int(e_b)))

    def set_act_exponents(self, sa_in, sa):
        arr = (C.c_int32 * len(sa))(*[int(v) for v in sa])
        _ffi.check(self._lib.y355_net_set_act_exponents(self._h, int(sa_in), arr, len(sa)))

    def get_act_exponents(self):
        sa_in = C.c_int32()
        arr = (C.c_int32 * self.num_tensors)()
        _ffi.check(self._lib.y355_net_get_act_exponents(self._h, C.byref(sa_in), arr, self.num_tensors))
        return sa_in.value, list(arr)

    def counters(self):
        s = C.c_int64()
        _ffi.check(self._lib.y355_net_counters(self._h, C.byref(s)))
        return s.value

    def set_option(self, option, value):
        """1 = Y355_NET_OPT_WORKGROUPS (throughput mode: persistent workgroups per ring launch, 0 = one per CU)"""
        _ffi.check(self._lib.y355_net_set_option(self._h, int(option), int(value)))

    def set_thresholds(self, conf_thresh, nms_thresh):
        _ffi.check(self._lib.y355_net_set_thresholds(self._h, float(conf_thresh), float(nms_thresh)))

    def _dev_input(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        if x.dim() != 4 or x.shape[1] != 3 or list(x.shape[2:]) != self.input_size:
            raise ValueError("expected [B,3,%d,%d], got %s" % (self.input_size[0], self.input_size[1], tuple(x.shape)))
        if x.shape[0] > self.max_batch:
            raise ValueError("batch %d > max_batch %d" % (x.shape[0], self.max_batch))
        return x.to(device=self.device, dtype=torch.float32).contiguous()

    def _buffers(self, B):
        if self._out is None:
            md = self.max_det
            self._out = (torch.empty((self.max_batch, md, 4), dtype=torch.float32, device=self.device),
                         torch.empty((self.max_batch, md), dtype=torch.float32, device=self.device),
                         torch.empty((self.max_batch, md), dtype=torch.int32, device=self.device),
                         torch.zeros((self.max_batch,), dtype=torch.int32, device=self.device))
        return self._out

    def forward_device(self, xd, flags=0, out=None):
        B = xd.shape[0]
        ob, os_, oc, on = out if out is not None else self._buffers(B)
        # stream ordering as in engine.Engine: kernels run on the stream current at construction
        cur = torch.cuda.current_stream(self.device)
        if cur != self._stream:
            self._stream.wait_stream(cur)
        _ffi.check(self._lib.y355_net_forward(self._h, xd.data_ptr(), B, int(flags), ob.data_ptr(), os_.data_ptr(),
                                              oc.data_ptr(), on.data_ptr()))
        if cur != self._stream:
            cur.wait_stream(self._stream)
        return ob, os_, oc, on

    def forward(self, x, tap=False):
        """list of (bboxes [n,4], scores [n], cls_inds int64 [n]) per image, anchor-index order."""
        xd = self._dev_input(x)
        B = xd.shape[0]
        ob, os_, oc, on = self.forward_device(xd, _ffi.F_TAP if tap else 0)
        n = on[:B].cpu().numpy()
        if self.overflow():
            raise _ffi.Y355Error(-1,
                                 "more than 4096 anchors of an image pass conf_thresh: raise the threshold")
        boxes, scores, cls = ob[:B].cpu().numpy(), os_[:B].cpu().numpy(), oc[:B].cpu().numpy()
        return [(boxes[i, :n[i]].copy(), scores[i, :n[i]].copy(), cls[i, :n[i]].astype(np.int64))
                for i in range(B)]

    def overflow(self):
        """True if a forward since the last call dropped candidates (heads with more than 4096 anchors per image)."""
        v = C.c_int(0)
        _ffi.check(self._lib.y355_net_overflow(self._h, C.byref(v)))
        return bool(v.value)

    def candidates(self, batch):
        N = self.num_anchors_total
        b = np.empty((batch, N, 4), np.float32)
        s = np.empty((batch, N), np.float32)
        c = np.empty((batch, N), np.int32)
        _ffi.check(self._lib.y355_net_get_candidates(self._h, batch, b.ctypes.data, s.ctypes.data, c.ctypes.data))
        return b, s, c

    def get_tensor(self, idx, batch):
        """activation tensor idx of the last forward as fp32 [B,C,H,W] (parity tap)."""
        c, hh, ww = self.tensor_shape(idx)
        out = np.empty((batch, c, hh, ww), np.float32)
        _ffi.check(self._lib.y355_net_get_tensor(self._h, idx, batch, out.ctypes.data))
        return out

    def tensor_absmax(self, idx, batch):
        m = C.c_float()
        _ffi.check(self._lib.y355_net_tensor_absmax(self._h, idx, batch, C.byref(m)))
        return float(m.value)

    def calibration_exponents(self, x):
        """bf16 nets: run x and return (sa_in, [sa per tensor]) = floor(log2(127 / max|.|)) of the network
        input and of every activation tensor -- the AveragedRangeTracker first-call rule
        (models/slim_yolo_v2.py:22-33) applied to this graph.  Feeds set_act_exponents of the int8 net
        (which overrides the entries of max-pool outputs with their inputs' exponents)."""
        from .prep import RangeTracker
        xd = self._dev_input(x)
        B = xd.shape[0]
        self.forward_device(xd, _ffi.F_TAP)      # tap forward: every tensor is written (the fused front end skips conv1's map)
        sa_in = RangeTracker().update(float(xd.abs().max().item()), True)
        sa = []
        for t in range(self.num_tensors):
            c, hh, ww = self.tensor_shape(t)
            if t >= self.num_tensors - (2 if self.arch == "tiny_yolo_v3" else 1):
                m = float(np.abs(self.get_tensor(t, B)).max())       # fp32 prediction maps
            else:
                m = self.tensor_absmax(t, B)
            sa.append(RangeTracker().update(m, True))
        return sa_in, sa

    def sync(self):
        _ffi.check(self._lib.y355_net_sync(self._h))

    def profile(self, enable=True):
        _ffi.check(self._lib.y355_net_profile(self._h, 1 if enable else 0))

    def profile_ms(self):
        n = self._lib.y355_net_num_timers(self._h)
        arr = (C.c_float * n)()
        _ffi.check(self._lib.y355_net_profile_get(self._h, arr))
        return list(arr)
