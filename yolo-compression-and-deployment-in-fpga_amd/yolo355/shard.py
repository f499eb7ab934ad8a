"""Batch sharding across the GPUs of one node.

Images are independent once the activation exponents are frozen (models/slim_yolo_v2.py:215,
28-29), so the path shards by batch with NO data-path collective: rank r runs images
[r*per_rank, (r+1)*per_rank) through its own engine.  The only exchange is one all-gather
of the fixed-cap padded detections per batch (RCCL over xGMI on GPUs; gloo in the CPU tests):
    boxes f32 [per_rank, max_det, 4], scores f32 [per_rank, max_det],
    cls i32 [per_rank, max_det], count i32 [per_rank]
Result order = global image index.  Calibration happens once (rank 0) and the 11 exponents
(44 bytes) are broadcast -- never per rank on different data.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(global_batch, world_size, rank):
    """contiguous shard of the global batch owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(global_batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_exponents(sa, src=0, device=None):
    """share rank `src`'s 11 activation exponents with every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [int(v) for v in sa]
    t = torch.tensor([int(v) for v in (sa if sa is not None else [0] * 11)], dtype=torch.int32, device=device)
    dist.broadcast(t, src)
    return [int(v) for v in t.cpu()]


def allgather_detections(boxes, scores, cls, count, async_op=False):
    """all-gather equal-shaped padded detection tensors; returns the gathered tensors
    (leading dim world*per_rank, global image order) and, if async_op, the work handles."""
    world = dist.get_world_size()
    outs, works = [], []
    for t in (boxes, scores, cls, count):
        g = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        w = dist.all_gather_into_tensor(g, t.contiguous(), async_op=async_op)
        outs.append(g)
        works.append(w)
    return (tuple(outs), works) if async_op else tuple(outs)


def unpack(boxes, scores, cls, count):
    """padded tensors -> list of (bboxes [n,4] f32, scores [n] f32, cls_inds [n] i64) numpy."""
    b, s, c, n = (t.cpu().numpy() for t in (boxes, scores, cls, count))
    return [(b[i, :n[i]].copy(), s[i, :n[i]].copy(), c[i, :n[i]].astype(np.int64)) for i in range(len(n))]
