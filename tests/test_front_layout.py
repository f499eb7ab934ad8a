"""Host-side check of the fused front end's operand layout (csrc/front.hip), no GPU needed.

The kernel computes the four conv outputs of a 2x2 pooling window as four MFMAs over ONE 4x4-neighbourhood operand
with four weight fragments (the 3x3 filter placed at the four offsets inside the neighbourhood).  Here the fragments
that `y355_pack_front_weights` produces are run through a numpy model of `v_mfma_i32_16x16x64_i8`
(A: lane l holds row l & 15, k = 16 (l >> 4) .. + 15; B: column l & 15, same k; D: column l & 15, rows 4 (l >> 4) + r)
with the kernel's own index arithmetic, and compared with a direct convolution + max-pool of
models/slim_yolo_v2.py:229-231 / :242-244 (integer restatement: SURVEY 8a-7)."""
import ctypes

import numpy as np

from yolo355 import _ffi


def _mfma(a_frag, b_frag, c):
    """a_frag, b_frag: int8 [64 lanes][16]; c: int32 [16 rows][16 cols] -> D = A B + C"""
    A = np.zeros((16, 64), np.int32)
    Bm = np.zeros((64, 16), np.int32)
    for l in range(64):
        A[l & 15, 16 * (l >> 4):16 * (l >> 4) + 16] = a_frag[l]
        Bm[16 * (l >> 4):16 * (l >> 4) + 16, l & 15] = b_frag[l]
    return A @ Bm + c


def _pack(w1, w2):
    dst = np.zeros(16384, np.int8)
    rc = _ffi.lib().y355_pack_front_weights(w1.ctypes.data_as(ctypes.c_void_p), w2.ctypes.data_as(ctypes.c_void_p),
                                            dst.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return dst.reshape(16, 64, 16)


def _conv_pool(x, w):
    """x int [C][H][W] (already padded as needed), w [O][C][3][3]: valid conv then 2x2 max-pool"""
    O, C = w.shape[:2]
    H, W = x.shape[1] - 2, x.shape[2] - 2
    y = np.zeros((O, H, W), np.int64)
    for ky in range(3):
        for kx in range(3):
            y += np.einsum("oc,chw->ohw", w[:, :, ky, kx].astype(np.int64), x[:, ky:ky + H, kx:kx + W].astype(np.int64))
    return y.reshape(O, H // 2, 2, W // 2, 2).max(axis=(2, 4))


def test_front_fragments_compute_conv_pool():
    rng = np.random.default_rng(7)
    w1 = rng.integers(-127, 128, (16, 3, 3, 3)).astype(np.int8)
    w2 = rng.integers(-127, 128, (32, 16, 3, 3)).astype(np.int8)
    fr = _pack(w1, w2)
    b1 = rng.integers(-2000, 2000, 16).astype(np.int32)
    b2 = rng.integers(-20000, 20000, 32).astype(np.int32)

    # ---- conv1: patch of 4-byte pixels (r, g, b, 0); 8 x 6 windows = 3 groups of 16
    WY, WX = 6, 8
    patch = rng.integers(-127, 128, (2 * WY + 2, 2 * WX + 2, 4)).astype(np.int8)
    patch[:, :, 3] = 0
    ref1 = _conv_pool(np.transpose(patch[:, :, :3], (2, 0, 1)), w1) + b1[:, None, None]
    got1 = np.zeros_like(ref1)
    for grp in range(WY * WX // 16):
        bfrag = np.zeros((64, 16), np.int8)
        for l in range(64):
            li, g = l & 15, l >> 4
            w = grp * 16 + li
            py, px = divmod(w, WX)
            bfrag[l] = patch[2 * py + g, 2 * px:2 * px + 4].reshape(16)
        cin = np.zeros((16, 16), np.int32)
        cin[:] = b1[:, None]                                   # lane (li, g), register r: bias of channel 4 g + r
        d = [_mfma(fr[v], bfrag, cin) for v in range(4)]
        pooled = np.maximum(np.maximum(d[0], d[1]), np.maximum(d[2], d[3]))
        for li in range(16):
            py, px = divmod(grp * 16 + li, WX)
            got1[:, py, px] = pooled[:, li]
    assert np.array_equal(got1, ref1)

    # ---- conv2: p1 of 16-byte pixels; 4 x 4 windows = one group; accumulator row i of n-tile n = channel 8 (i >> 2) + 4 n + (i & 3)
    p1 = rng.integers(-127, 128, (10, 10, 16)).astype(np.int8)
    ref2 = _conv_pool(np.transpose(p1, (2, 0, 1)), w2) + b2[:, None, None]
    got2 = np.zeros_like(ref2)
    acc = {}
    for dy in range(2):
        for dx in range(2):
            for n in range(2):
                chan = [8 * (i >> 2) + 4 * n + (i & 3) for i in range(16)]
                c = np.zeros((16, 16), np.int32)
                c[:] = b2[chan][:, None]
                acc[dy, dx, n] = c
    for t in range(4):
        bfrag = np.zeros((64, 16), np.int8)
        for l in range(64):
            li, g = l & 15, l >> 4
            wy, wx = divmod(li, 4)
            bfrag[l] = p1[2 * wy + t, 2 * wx + g]
        for dy in range(2):
            ky = t - dy
            if ky < 0 or ky > 2:
                continue
            for dx in range(2):
                for n in range(2):
                    acc[dy, dx, n] = _mfma(fr[4 + (n * 3 + ky) * 2 + dx], bfrag, acc[dy, dx, n])
    for n in range(2):
        pooled = np.maximum(np.maximum(acc[0, 0, n], acc[0, 1, n]), np.maximum(acc[1, 0, n], acc[1, 1, n]))
        for i in range(16):
            ch = 8 * (i >> 2) + 4 * n + (i & 3)
            for li in range(16):
                wy, wx = divmod(li, 4)
                got2[ch, wy, wx] = pooled[i, li]
    assert np.array_equal(got2, ref2)


def test_front_fp32_requant_is_the_integer_pipeline():
    """q = low byte of med3(max(fma(t, 2^(lk-sh), M), fma(t, 2^-sh, M)), M -+ 127) against clamp(RNE(max(t, 8 t) * 2^-sh)),
    and the biased form fma(M + t, s, M (1 - s)) the kernel uses when the bias rides in the accumulator: exhaustive around
    the rounding ties and the clamp, random elsewhere (float32 arithmetic emulated with exactly-rounded float64 -> float32:
    every product is a power-of-two scaling and every sum is rounded once, as the fma does)."""
    M = np.float32(12582912.0)
    rng = np.random.default_rng(3)
    for sh in (1, 4, 9, 11, 14, 22):
        s_pos, s_neg = np.float64(2.0 ** (3 - sh)), np.float64(2.0 ** -sh)
        t = np.concatenate([rng.integers(-2 ** 21, 2 ** 21, 200000), np.arange(-5000, 5000),
                            (np.arange(-140, 141)[:, None] * 2 ** sh // 8 + np.arange(-3, 4)[None, :]).ravel(),
                            (np.arange(-140, 141)[:, None] * 2 ** sh + 2 ** (sh - 1) + np.arange(-3, 4)[None, :]).ravel()]).astype(np.int64)
        t = t[np.abs(t) < 2 ** 22]
        tp = np.maximum(t, 8 * t)
        q_ref = np.clip(np.rint(tp.astype(np.float64) * 2.0 ** -sh), -127, 127).astype(np.int64)   # rint = half to even; exact in f64
        tf = t.astype(np.float32)
        y = np.maximum((tf.astype(np.float64) * s_pos + np.float64(M)).astype(np.float32),
                       (tf.astype(np.float64) * s_neg + np.float64(M)).astype(np.float32))
        yc = np.minimum(np.maximum(y, np.float32(M - 127)), np.float32(M + 127))
        q = (yc.view(np.uint32) & 0xff).astype(np.uint8).view(np.int8).astype(np.int64)
        assert np.array_equal(q, q_ref), sh
        tb = (t + 0x4B400000).astype(np.uint32).view(np.float32)                       # accumulator bits read as a float: M + t
        assert np.array_equal(tb.astype(np.float64), np.float64(M) + t)
        c_pos, c_neg = np.float32(M) - np.float32(M) * np.float32(s_pos), np.float32(M) - np.float32(M) * np.float32(s_neg)
        assert float(c_pos) == float(M) * (1 - float(s_pos)) and float(c_neg) == float(M) * (1 - float(s_neg))
        yb = np.maximum((tb.astype(np.float64) * s_pos + np.float64(c_pos)).astype(np.float32),
                        (tb.astype(np.float64) * s_neg + np.float64(c_neg)).astype(np.float32))
        assert np.array_equal(yb, y), sh


def test_ring_fp32_requant_covers_the_whole_int32_range():
    """conv3x3_ring.hip's FPE epilogue (conv5 .. pred): t = acc + bias may be ANY int32 -- the launcher only proves that the
    right shift is at most 17 bits.  Then every t that does not saturate is below 2^24 (exact in fp32) and every larger one
    converts (round to nearest even) to something at least as large and saturates either way: the byte AND the 'this value
    was clamped' flag equal the integer pipeline's for leaky (max(t, 8 t)) and linear (pred) layers, at and around 2^24,
    the clamp boundary, the rounding ties and the int32 limits."""
    M = np.float32(12582912.0)
    lo, hi = np.float32(M - 127), np.float32(M + 127)
    rng = np.random.default_rng(11)
    edge = np.concatenate([np.arange(-4, 5) + s * 2 ** k for k in range(20, 32) for s in (-1, 1)]
                          + [np.array([-2 ** 31, -2 ** 31 + 1, 2 ** 31 - 1, 2 ** 31 - 2])])
    for lk in (0, 3):
        for sh in range(0, 18):
            t = np.concatenate([rng.integers(-2 ** 31, 2 ** 31, 100000), rng.integers(-2 ** 25, 2 ** 25, 100000),
                                rng.integers(-130 << sh, (130 << sh) + 1, 50000), edge,
                                (np.arange(-140, 141)[:, None] * 2 ** sh + (2 ** sh >> 1) + np.arange(-3, 4)[None, :]).ravel(),
                                (np.arange(-140, 141)[:, None] * (2 ** sh) // (2 ** lk) + np.arange(-3, 4)[None, :]).ravel()]).astype(np.int64)
            t = t[(t >= -2 ** 31) & (t < 2 ** 31)]
            tp = np.maximum(t, t * 2 ** lk)                                            # exact in int64
            half = (1 << (sh - 1)) - 1 if sh > 0 else 0
            qq = (tp + half + ((tp >> sh) & (1 if sh > 0 else 0))) >> sh               # the integer pipeline: RNE shift
            q_ref, sat_ref = np.clip(qq, -127, 127), np.abs(qq) > 127
            tf = t.astype(np.int32).astype(np.float32)                                 # v_cvt_f32_i32: round to nearest even
            y = np.maximum((tf.astype(np.float64) * 2.0 ** (lk - sh) + np.float64(M)).astype(np.float32),
                           (tf.astype(np.float64) * 2.0 ** -sh + np.float64(M)).astype(np.float32))
            yc = np.minimum(np.maximum(y, lo), hi)
            q = (yc.view(np.uint32) & 0xff).astype(np.uint8).view(np.int8).astype(np.int64)
            assert np.array_equal(q, q_ref), (lk, sh)
            assert np.array_equal(y != yc, sat_ref), (lk, sh)
    # and the bound is tight: one more bit of shift and an odd t just above 2^24 lands on a tie it is not on
    sh, t = 18, np.array([2 ** 24 + 2 ** 17 + 1], np.int64)
    qq = (t + (1 << (sh - 1)) - 1 + ((t >> sh) & 1)) >> sh
    tf = t.astype(np.int32).astype(np.float32)
    y = (tf.astype(np.float64) * 2.0 ** -sh + np.float64(M)).astype(np.float32)
    assert int(y.view(np.uint32)[0] & 0xff) != int(qq[0])
