"""cv2.resize(image, (W, H)) as BaseTransform calls it (data/__init__.py:36): 8-bit, 3 channels, INTER_LINEAR (cv2's default).
TEST INFRASTRUCTURE ONLY (checker of y355_forward_u8's resize stage).

Parity status: UNPINNED.  OpenCV is a third-party dependency of the reference that is neither vendored in /root/reference
nor installed in the build image (requirement `opencv-python`, no version pinned by the reference's README), so no golden
vector can be produced here.  This restates the published algorithm of OpenCV 4.x `imgproc/src/resize.cpp` for CV_8U:
  * coordinates: fx = float((dx + 0.5) * scale_x - 0.5), sx = floor(fx), fx -= sx; sx < 0 -> (0, fx = 0);
    sx >= src_w - 1 -> (src_w - 1, fx = 0); scale = src / dst in double.  In y the offset and the fraction are kept as they
    are and the two row indices are clipped instead (resizeGeneric_Invoker: sy = clip(sy0 - ksize2 + 1 + k, 0, src_h)): a
    border row is blended with itself through two separately truncated products
  * fixed-point coefficients: short(round_half_even(c * 2048)) for c in (1 - f, f)   (INTER_RESIZE_COEF_BITS = 11)
  * horizontal pass in int32:  D = S[sx] * a0 + S[sx + 1] * a1
  * vertical pass (VResizeLinear<uchar>): dst = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2
  * equal sizes reproduce the input; an exact 2x decimation equals the (a + b + c + d + 2) >> 2 average OpenCV switches to.
"""
import numpy as np


def linear_tables(src, dst, vertical=False):
    """(ofs int32 [dst], coef int16 [dst, 2]) of one axis (vertical: unclamped offsets, see the header)."""
    scale = float(src) / float(dst)
    ofs = np.zeros(dst, np.int32)
    coef = np.zeros((dst, 2), np.int32)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if not vertical and s < 0:
            s, f = 0, np.float32(0.0)
        if not vertical and s >= src - 1:
            s, f = src - 1, np.float32(0.0)
        ofs[d] = s
        c0 = np.float32(np.float32(1.0) - f) * np.float32(2048.0)
        c1 = f * np.float32(2048.0)
        coef[d, 0] = int(np.clip(np.rint(c0), -32768, 32767))
        coef[d, 1] = int(np.clip(np.rint(c1), -32768, 32767))
    return ofs, coef


def resize_linear_u8(img, dst_h, dst_w):
    """img uint8 [..., h, w, C] -> uint8 [..., dst_h, dst_w, C]"""
    img = np.asarray(img, np.uint8)
    h, w = img.shape[-3], img.shape[-2]
    xo, xa = linear_tables(w, dst_w)
    yo, yb = linear_tables(h, dst_h, vertical=True)
    x1 = np.minimum(xo + 1, w - 1)
    y1 = np.clip(yo + 1, 0, h - 1)
    yo = np.clip(yo, 0, h - 1)
    s = img.astype(np.int32)
    hz = s[..., :, xo, :] * xa[:, 0][:, None] + s[..., :, x1, :] * xa[:, 1][:, None]           # [..., h, dst_w, C]
    d0 = hz[..., yo, :, :] >> 4
    d1 = hz[..., y1, :, :] >> 4
    b0 = yb[:, 0][:, None, None]
    b1 = yb[:, 1][:, None, None]
    out = (((b0 * d0) >> 16) + ((b1 * d1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)
