"""Shared comparison helpers for the parity tests."""
import zlib

import numpy as np


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def dets_match(ref, got, box_tol=2e-5, score_tol=2e-6, all_scores=None):
    """Compare two detection lists given in anchor-index order.

    Strict: same count, same classes, boxes/scores within tolerance.
    The reference's NMS order inside groups of exactly equal scores is undefined
    (unstable argsort()[::-1], models/slim_yolo_v2.py:153) and int8 logits
    make such groups large.  When the strict comparison fails we fall back to a set
    comparison in which every detection present on one side only must have a score that
    is shared by at least two candidates of the image (`all_scores` = per-anchor best-class
    scores of that image; defaults to the union of both lists).  Returns (ok, message)."""
    rb, rs, rc = ref
    gb, gs, gc = got
    if len(rs) == len(gs):
        if len(rs) == 0:
            return True, "empty"
        if (np.array_equal(rc, gc) and np.allclose(rb, gb, atol=box_tol, rtol=0)
                and np.allclose(rs, gs, atol=score_tol, rtol=1e-5)):
            return True, "exact"
    q = max(box_tol, 1e-6)

    def keyset(b, s, c):
        # boxes clamped to the image border can coincide: the k-th occurrence (anchor order)
        # of a (box, class) pair is its own key
        out, seen = {}, {}
        for bb, ss, cc in zip(b, s, c):
            k0 = (tuple(np.round(bb / q).astype(np.int64)), int(cc))
            n = seen.get(k0, 0)
            seen[k0] = n + 1
            out[k0 + (n,)] = float(ss)
        return out
    R, G = keyset(rb, rs, rc), keyset(gb, gs, gc)
    only = sorted([R[k] for k in R if k not in G] + [G[k] for k in G if k not in R], reverse=True)
    if not only:
        # same boxes and classes on both sides: the strict test failed on a score or a count
        if len(R) != len(rs) or len(G) != len(gs):
            return False, "duplicate boxes: ref %d/%d got %d/%d" % (len(R), len(rs), len(G), len(gs))
        worst = max(abs(R[k] - G[k]) for k in R)
        ok = all(abs(R[k] - G[k]) <= score_tol + 1e-5 * abs(R[k]) for k in R)
        return ok, "same boxes, max score diff %.3g" % worst
    pool = np.sort(np.asarray(all_scores if all_scores is not None else np.concatenate([rs, gs]),
                              dtype=np.float64))
    tol = max(score_tol, 1e-9)
    # Greedy NMS is sequential in score order: once two runs order one tied group
    # differently, every lower-scoring decision may legitimately differ (cascade).  So the
    # HIGHEST-scoring one-sided detection must sit in a tie group; everything above it
    # matched exactly by construction.
    top = only[0]
    lo = np.searchsorted(pool, top - tol, "left")
    hi = np.searchsorted(pool, top + tol, "right")
    if hi - lo < 2:
        return False, "first divergence at untied score %r (ref %d, got %d, %d one-sided)" % (
            top, len(rs), len(gs), len(only))
    if abs(len(rs) - len(gs)) > max(4, 0.02 * max(len(rs), len(gs))):
        return False, "count differs beyond tie slack: ref %d got %d" % (len(rs), len(gs))
    return True, "tie-tolerant: %d one-sided below tied score %.6g" % (len(only), top)


def iou_matrix(a, b):
    """pairwise IoU of x1y1x2y2 boxes a [n,4], b [m,4] (float64)."""
    a = np.asarray(a, np.float64)[:, None, :]
    b = np.asarray(b, np.float64)[None, :, :]
    iw = np.clip(np.minimum(a[..., 2], b[..., 2]) - np.maximum(a[..., 0], b[..., 0]), 0, None)
    ih = np.clip(np.minimum(a[..., 3], b[..., 3]) - np.maximum(a[..., 1], b[..., 1]), 0, None)
    inter = iw * ih
    ua = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]) + (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]) - inter
    return inter / np.maximum(ua, 1e-12)


def dets_close(ref, got, iou_min=0.8, score_tol=0.05):
    """Tolerance comparison of two detection lists (floating-point paths): fraction of `ref`
    detections that have a `got` detection of the same class with IoU >= iou_min and a score within
    score_tol, and vice versa.  Returns (frac_ref_matched, frac_got_matched)."""
    rb, rs, rc = ref[:3]
    gb, gs, gc = got[:3]
    if len(rs) == 0 or len(gs) == 0:
        return (1.0 if len(rs) == 0 else 0.0), (1.0 if len(gs) == 0 else 0.0)
    iou = iou_matrix(rb, gb)
    ok = (iou >= iou_min) & (np.asarray(rc)[:, None] == np.asarray(gc)[None, :]) & \
         (np.abs(np.asarray(rs, np.float64)[:, None] - np.asarray(gs, np.float64)[None, :]) <= score_tol)
    return float(ok.any(axis=1).mean()), float(ok.any(axis=0).mean())
