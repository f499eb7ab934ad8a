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


# ---------------------------------------------------------------------------------------------------------------------
# Attribution of detection-LIST differences between a floating-point reference and the engine.
# The per-anchor decode of the bf16 / int8 paths differs from the fp32 reference's within stated tolerances (scores, boxes);
# greedy NMS then amplifies marginal differences: an anchor a hair on the other side of conf_thresh, a pair whose IoU sits
# on the other side of nms_thresh, and everything those decisions suppress or release downstream.  A detection-level match
# rate (dets_close) cannot tell that apart from a wrong NMS.  This replays both sides' per-class greedy NMS on the PER-ANCHOR
# candidates (same anchor index on both sides) and demands a cause for every anchor whose fate differs:
#   thr      the anchor's best score is on different sides of conf_thresh on the two sides
#   cls      the anchor's best class differs (per-class NMS: it competes in another list)
#   iou      on the side where it is suppressed, its suppressor s is kept on BOTH sides, but IoU(a, s) is on different
#            sides of nms_thresh (or the score order of a and s is)
#   cascade  its suppressor's own fate differs between the sides (s is then classified itself: chains end in a root cause)
# Anything else is `unexplained` -- e.g. an anchor dropped with no kept, higher-scored, overlapping neighbour on that side --
# and means the NMS itself differs.  The root causes' per-anchor deviations are returned so that the caller can hold them to
# the per-anchor tolerances.
def _nms_replay(box, score, cls, conf_thresh, nms_thresh):
    """Greedy per-class NMS of models/slim_yolo_v2.py:145-210 on per-anchor candidates (order: score descending, anchor
    ascending).  Returns (state int8 [N]: 0 below conf_thresh, 1 kept, 2 suppressed; suppressor int [N] or -1)."""
    box = np.asarray(box, np.float32)
    score = np.asarray(score, np.float32)
    n = len(score)
    state = np.zeros(n, np.int8)
    sup = np.full(n, -1, np.int64)
    cand = np.where(score >= np.float32(conf_thresh))[0]
    area = (box[:, 2] - box[:, 0]) * (box[:, 3] - box[:, 1])
    for c in np.unique(cls[cand]):
        idx = cand[cls[cand] == c]
        idx = idx[np.lexsort((idx, -score[idx].astype(np.float64)))]
        kept = []
        for a in idx:
            if kept:
                k = np.asarray(kept)
                xx1 = np.maximum(box[a, 0], box[k, 0]); yy1 = np.maximum(box[a, 1], box[k, 1])
                xx2 = np.minimum(box[a, 2], box[k, 2]); yy2 = np.minimum(box[a, 3], box[k, 3])
                w = np.maximum(np.float32(1e-28), xx2 - xx1); h = np.maximum(np.float32(1e-28), yy2 - yy1)
                inter = w * h
                iou = inter / (area[a] + area[k] - inter)
                hit = np.where(iou > np.float32(nms_thresh))[0]
                if len(hit):
                    state[a], sup[a] = 2, k[hit[0]]          # the first (highest-scored) kept box that suppresses it
                    continue
            state[a] = 1
            kept.append(a)
    return state, sup


def _iou1(b, a, s):
    xx1 = max(b[a, 0], b[s, 0]); yy1 = max(b[a, 1], b[s, 1]); xx2 = min(b[a, 2], b[s, 2]); yy2 = min(b[a, 3], b[s, 3])
    w = max(1e-28, float(xx2 - xx1)); h = max(1e-28, float(yy2 - yy1))
    ar = lambda i: float(b[i, 2] - b[i, 0]) * float(b[i, 3] - b[i, 1])
    return w * h / (ar(a) + ar(s) - w * h)


def explain_detection_differences(ref_box, ref_score, ref_cls, got_box, got_score, got_cls, conf_thresh, nms_thresh):
    """Per-anchor candidates of both sides ([N,4], [N], [N]) -> dict(n_ref, n_got, differ, causes={thr, cls, iou, cascade},
    unexplained=[anchor, ...], root_score_dev (max |score difference| over the thr / cls roots), root_box_dev (max
    |coordinate difference| over the anchors of the iou roots), kept_ref, kept_got)."""
    ref_box, got_box = np.asarray(ref_box, np.float32), np.asarray(got_box, np.float32)
    ref_score, got_score = np.asarray(ref_score, np.float32), np.asarray(got_score, np.float32)
    ref_cls, got_cls = np.asarray(ref_cls).astype(np.int64), np.asarray(got_cls).astype(np.int64)
    sides = [(ref_box, ref_score, ref_cls) + _nms_replay(ref_box, ref_score, ref_cls, conf_thresh, nms_thresh),
             (got_box, got_score, got_cls) + _nms_replay(got_box, got_score, got_cls, conf_thresh, nms_thresh)]
    st = [sides[0][3], sides[1][3]]
    differ = np.where((st[0] == 1) != (st[1] == 1))[0]
    causes = dict(thr=0, cls=0, iou=0, cascade=0)
    unexplained, sdev, bdev = [], 0.0, 0.0
    dset = set(int(a) for a in differ)
    for a in differ:
        a = int(a)
        if (st[0][a] == 0) != (st[1][a] == 0):
            causes["thr"] += 1
            sdev = max(sdev, abs(float(ref_score[a]) - float(got_score[a])))
            continue
        if ref_cls[a] != got_cls[a]:
            causes["cls"] += 1
            sdev = max(sdev, abs(float(ref_score[a]) - float(got_score[a])))
            continue
        x = 0 if st[0][a] == 2 else 1                     # the side on which it is suppressed; kept on the other
        y = 1 - x
        if st[x][a] != 2 or st[y][a] != 1:
            unexplained.append(a)
            continue
        s = int(sides[x][4][a])
        if s in dset or st[y][s] != 1:
            causes["cascade"] += 1                         # its suppressor's own fate differs: classified on its own
            continue
        by, sy, cy = sides[y][0], sides[y][1], sides[y][2]
        order_flip = not (sy[s] > sy[a] or (sy[s] == sy[a] and s < a))
        if cy[s] != cy[a] or _iou1(by, a, s) <= nms_thresh or order_flip:
            causes["iou"] += 1
            bdev = max(bdev, float(np.abs(ref_box[[a, s]] - got_box[[a, s]]).max()))
            continue
        unexplained.append(a)                              # kept beside a kept, higher-scored, overlapping box of its class
    return dict(n_ref=int((st[0] == 1).sum()), n_got=int((st[1] == 1).sum()), differ=int(len(differ)), causes=causes,
                unexplained=unexplained, root_score_dev=sdev, root_box_dev=bdev,
                kept_ref=np.where(st[0] == 1)[0], kept_got=np.where(st[1] == 1)[0])
