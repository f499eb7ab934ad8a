"""ctypes loader of oracle/_build/libyolo_oracle.so (checker / CPU baseline only)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libyolo_oracle.so")


class Layer(C.Structure):
    _fields_ = [("cout", C.c_int32), ("cin", C.c_int32), ("e_w", C.c_int32), ("e_b", C.c_int32),
                ("q_w", C.c_void_p), ("q_b", C.c_void_p)]


def lib():
    if not os.path.exists(_LIB):
        raise ImportError("build the C oracle first: make -C oracle")
    return C.CDLL(_LIB)


def detect(x, qlayers, sa, input_size, anchors, num_classes, conf_thresh=0.01, nms_thresh=0.5, saturate=True):
    """x fp32 [B,3,H,W]; returns (pred_q int8 [B,PC,Hs,Ws], nsat[11], dets list)."""
    L = lib()
    x = np.ascontiguousarray(x, np.float32)
    B, _, H, W = x.shape
    keep = []
    arr = (Layer * 10)()
    for i, q in enumerate(qlayers):
        qw = np.ascontiguousarray(q["q_w"], np.int8)
        qb = np.ascontiguousarray(q["q_b"], np.int32)
        keep += [qw, qb]
        arr[i] = Layer(qw.shape[0], qw.shape[1], int(q["e_w"]), int(q["e_b"]), qw.ctypes.data, qb.ctypes.data)
    A = len(anchors)
    PC = A * (5 + num_classes)
    Hs, Ws = H // 16, W // 16
    pred = np.zeros((B, PC, Hs, Ws), np.int8)
    nsat = np.zeros(11, np.int64)
    sa_arr = np.asarray(sa, np.int32)
    rc = L.yo_backbone(x.ctypes.data_as(C.c_void_p), B, H, W, arr, sa_arr.ctypes.data_as(C.c_void_p), int(saturate),
                       pred.ctypes.data_as(C.c_void_p), nsat.ctypes.data_as(C.c_void_p))
    assert rc == 0
    N = Hs * Ws * A
    boxes = np.zeros((B, N, 4), np.float32)
    scores = np.zeros((B, N), np.float32)
    cls = np.zeros((B, N), np.int32)
    count = np.zeros(B, np.int32)
    anc = np.ascontiguousarray(np.asarray(anchors, np.float32).reshape(-1))
    L.yo_head_nms(pred.ctypes.data_as(C.c_void_p), B, Hs, Ws, A, num_classes, anc.ctypes.data_as(C.c_void_p),
                  int(sa[10]), C.c_float(conf_thresh), C.c_float(nms_thresh), int(input_size[0]), int(input_size[1]),
                  N, boxes.ctypes.data_as(C.c_void_p), scores.ctypes.data_as(C.c_void_p),
                  cls.ctypes.data_as(C.c_void_p), count.ctypes.data_as(C.c_void_p))
    dets = [(boxes[i, :count[i]].copy(), scores[i, :count[i]].copy(), cls[i, :count[i]].astype(np.int64)) for i in range(B)]
    return pred, nsat, dets
