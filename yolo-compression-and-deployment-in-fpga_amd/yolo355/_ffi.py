"""ctypes binding of include/yolo355.h.  Fails loudly when libyolo355.so is absent."""
import ctypes as C
import os

# torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  It must be in the process
# BEFORE libyolo355.so is loaded so that the loader binds both to ONE HIP runtime; loaded the
# other way round the process ends up with two runtimes and the second sees no device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libyolo355.so")
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "yolo355.h"))

MAX_ANCHORS = 16
NUM_TIMERS = 12
NUM_KERNEL_TIMERS = 14
F_GUARD, F_TAP = 1, 2
PIPE_AFTER_STREAM = 0x100
PIPE_DEFAULT_HANDLES = 4
OP_LEAKY, OP_POOL, OP_RELU = 1, 2, 4
OPT_FUSE_FRONT = 1
OPT_RING_WORKGROUPS = 2
OPT_FUSE_PAIRS = 3
NET_OPT_WORKGROUPS = 1
NET_OPT_THIN_RESIDENT = 2
EINVAL, EHIP, ENOTREADY = -1, -2, -3


class Config(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
                ("num_classes", C.c_int32), ("num_anchors", C.c_int32),
                ("anchors", C.c_float * (2 * MAX_ANCHORS)),
                ("conf_thresh", C.c_float), ("nms_thresh", C.c_float),
                ("max_batch", C.c_int32), ("max_det", C.c_int32), ("stream", C.c_void_p),
                ("own_stream", C.c_int32)]


class NetConfig(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("arch", C.c_int32), ("dtype", C.c_int32),
                ("height", C.c_int32), ("width", C.c_int32),
                ("num_classes", C.c_int32), ("num_anchors", C.c_int32),
                ("anchors", C.c_float * (2 * MAX_ANCHORS)),
                ("conf_thresh", C.c_float), ("nms_thresh", C.c_float),
                ("max_batch", C.c_int32), ("max_det", C.c_int32), ("stream", C.c_void_p),
                ("own_stream", C.c_int32)]


ARCH_SLIM_V2, ARCH_TINY_V3, ARCH_YOLO_V2, ARCH_YOLO_V3, ARCH_YOLO_V3_SPP = 0, 1, 2, 3, 4
DT_INT8, DT_BF16 = 0, 1


class LayerStats(C.Structure):
    _fields_ = [("absmax_t", C.c_int64), ("frac_bits", C.c_int32), ("reserved", C.c_int32),
                ("saturated", C.c_int64), ("guard", C.c_int64)]


class Y355Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("yolo355 error %d: %s" % (code, msg))
        self.code = code


_lib = None
P = C.POINTER
_SIGS = {
    "y355_last_error": (C.c_char_p, []),
    "y355_version": (C.c_int, []),
    "y355_create": (C.c_int, [P(Config), P(C.c_void_p)]),
    "y355_destroy": (None, [C.c_void_p]),
    "y355_set_thresholds": (C.c_int, [C.c_void_p, C.c_float, C.c_float]),
    "y355_pack_front_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "y355_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "y355_load_layer": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "y355_set_act_exponents": (C.c_int, [C.c_void_p, P(C.c_int32)]),
    "y355_get_act_exponents": (C.c_int, [C.c_void_p, P(C.c_int32)]),
    "y355_set_act_exponent": (C.c_int, [C.c_void_p, C.c_int, C.c_int32]),
    "y355_set_retune": (C.c_int, [C.c_void_p, P(C.c_int32)]),
    "y355_input_absmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, P(C.c_float)]),
    "y355_run_layer": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y355_layer_stats_get": (C.c_int, [C.c_void_p, C.c_int, P(LayerStats)]),
    "y355_get_feature": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "y355_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "y355_set_normalization": (C.c_int, [C.c_void_p, P(C.c_float), P(C.c_float)]),
    "y355_forward_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "y355_forward_u8_resized": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "y355_forward_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "y355_forward_counters": (C.c_int, [C.c_void_p, P(C.c_int64), P(C.c_int64)]),
    "y355_get_candidates": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "y355_scale_boxes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "y355_max_det": (C.c_int, [C.c_void_p]),
    "y355_num_anchors_total": (C.c_int, [C.c_void_p]),
    "y355_conv3x3_i8_fused": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, P(LayerStats)]),
    "y355_conv3x3_i8_raw": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, P(C.c_int32)]),
    "y355_quantize_input_f32_i8": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, P(C.c_int64)]),
    "y355_maxpool2x2_i8": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y355_reorg_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y355_spp_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y355_conv2d_bf16": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "y355_maxpool2x2_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y355_upsample2x_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y355_reorg_f32_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "y355_spp_f32_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "y355_maxpool2x2_f32_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "y355_upsample2x_f32_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "y355_conv_op_create_bf16": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, P(C.c_void_p)]),
    "y355_conv_op_create_i8": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_void_p)]),
    "y355_conv_op_destroy": (None, [C.c_void_p]),
    "y355_conv_op_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "y355_conv_op_forward_i8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, P(C.c_int32), P(C.c_int32)]),
    "y355_head_f32": (C.c_int, [C.c_int, C.c_int, P(C.c_void_p), P(C.c_int), P(C.c_int), P(C.c_float), P(C.c_float), C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
    "y355_debug_stamps": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "y355_debug_nms_stamps": (C.c_int, [C.c_void_p]),
    "y355_head_nms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "y355_packed_det_bytes": (C.c_size_t, [C.c_int]),
    "y355_comm_unique_id": (C.c_int, [C.c_void_p]),
    "y355_comm_init": (C.c_int, [P(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "y355_comm_destroy": (None, [C.c_void_p]),
    "y355_comm_world": (C.c_int, [C.c_void_p]),
    "y355_comm_rank": (C.c_int, [C.c_void_p]),
    "y355_pack_dets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "y355_pack_dets_capped": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "y355_allgather_dets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "y355_unpack_dets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "y355_set_trackers": (C.c_int, [C.c_void_p, P(C.c_float), P(C.c_int32)]),
    "y355_get_trackers": (C.c_int, [C.c_void_p, P(C.c_float), P(C.c_int32)]),
    "y355_calibrate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, P(C.c_int32), P(C.c_float)]),
    "y355_tracker_step": (C.c_int, [P(C.c_float), P(C.c_int32), C.c_float, C.c_int, C.c_double, P(C.c_int32)]),
    "y355_pipeline_create": (C.c_int, [P(Config), C.c_int, C.c_int, P(C.c_void_p)]),
    "y355_pipeline_create_on": (C.c_int, [P(Config), C.c_int, C.c_int, P(C.c_void_p), P(C.c_void_p)]),
    "y355_pipeline_destroy": (None, [C.c_void_p]),
    "y355_pipeline_handles": (C.c_int, [C.c_void_p]),
    "y355_pipeline_depth": (C.c_int, [C.c_void_p]),
    "y355_pipeline_max_det": (C.c_int, [C.c_void_p]),
    "y355_pipeline_engine": (C.c_void_p, [C.c_void_p, C.c_int]),
    "y355_pipeline_stream": (C.c_void_p, [C.c_void_p, C.c_longlong]),
    "y355_pipeline_load_layer": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "y355_pipeline_set_act_exponents": (C.c_int, [C.c_void_p, P(C.c_int32)]),
    "y355_pipeline_set_retune": (C.c_int, [C.c_void_p, P(C.c_int32)]),
    "y355_pipeline_set_thresholds": (C.c_int, [C.c_void_p, C.c_float, C.c_float]),
    "y355_pipeline_set_normalization": (C.c_int, [C.c_void_p, P(C.c_float), P(C.c_float)]),
    "y355_pipeline_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "y355_pipeline_set_trackers": (C.c_int, [C.c_void_p, P(C.c_float), P(C.c_int32)]),
    "y355_pipeline_get_trackers": (C.c_int, [C.c_void_p, P(C.c_float), P(C.c_int32)]),
    "y355_pipeline_calibrate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, P(C.c_int32), P(C.c_float)]),
    "y355_pipeline_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, P(C.c_longlong)]),
    "y355_pipeline_submit_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, P(C.c_longlong)]),
    "y355_pipeline_wait": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]),
    "y355_pipeline_outputs": (C.c_int, [C.c_void_p, C.c_longlong, P(C.c_void_p), P(C.c_void_p), P(C.c_void_p), P(C.c_void_p), P(C.c_int)]),
    "y355_pipeline_release": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p]),
    "y355_pipeline_fetch": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "y355_pipeline_scale_boxes": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p]),
    "y355_pipeline_counters": (C.c_int, [C.c_void_p, P(C.c_int64), P(C.c_int64)]),
    "y355_pipeline_sync": (C.c_int, [C.c_void_p]),
    "y355_stream": (C.c_void_p, [C.c_void_p]),
    "y355_sync": (C.c_int, [C.c_void_p]),
    "y355_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "y355_profile_get": (C.c_int, [C.c_void_p, P(C.c_float)]),
    "y355_profile_kernel_get": (C.c_int, [C.c_void_p, P(C.c_float)]),
    "y355_profile_kernels_get": (C.c_int, [C.c_void_p, P(C.c_float)]),
    "y355_mfma_peak_i8": (C.c_int, [C.c_int, C.c_float, P(C.c_float), P(C.c_float)]),
    "y355_net_create": (C.c_int, [P(NetConfig), P(C.c_void_p)]),
    "y355_net_destroy": (None, [C.c_void_p]),
    "y355_net_set_thresholds": (C.c_int, [C.c_void_p, C.c_float, C.c_float]),
    "y355_net_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "y355_net_num_layers": (C.c_int, [C.c_void_p]),
    "y355_net_num_tensors": (C.c_int, [C.c_void_p]),
    "y355_net_layer_shape": (C.c_int, [C.c_void_p, C.c_int, P(C.c_int32)]),
    "y355_net_tensor_shape": (C.c_int, [C.c_void_p, C.c_int, P(C.c_int32)]),
    "y355_net_load_layer_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "y355_net_load_layer_i8": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "y355_net_set_act_exponents": (C.c_int, [C.c_void_p, C.c_int, P(C.c_int32), C.c_int]),
    "y355_net_get_act_exponents": (C.c_int, [C.c_void_p, P(C.c_int32), P(C.c_int32), C.c_int]),
    "y355_net_counters": (C.c_int, [C.c_void_p, P(C.c_int64)]),
    "y355_net_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "y355_net_get_candidates": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "y355_net_get_tensor": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "y355_net_tensor_absmax": (C.c_int, [C.c_void_p, C.c_int, C.c_int, P(C.c_float)]),
    "y355_net_max_det": (C.c_int, [C.c_void_p]),
    "y355_net_num_anchors_total": (C.c_int, [C.c_void_p]),
    "y355_net_sync": (C.c_int, [C.c_void_p]),
    "y355_net_debug_nms": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "y355_net_overflow": (C.c_int, [C.c_void_p, P(C.c_int)]),
    "y355_net_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "y355_net_num_timers": (C.c_int, [C.c_void_p]),
    "y355_net_profile_get": (C.c_int, [C.c_void_p, P(C.c_float)]),
}


def lib():
    """Load libyolo355.so (built by `make -C csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "yolo355: %s is missing -- build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def check(rc):
    if rc != 0:
        raise Y355Error(rc, lib().y355_last_error().decode("utf-8", "replace"))
    return rc


def declared_symbols():
    """Function names declared in include/yolo355.h (used by the export test)."""
    import re
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(y355_[a-z0-9_]+)\s*\(", text)))
