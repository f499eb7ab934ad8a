"""Batch sharding across the GPUs of one node.

Images are independent once the activation exponents are frozen (models/slim_yolo_v2.py:215,
28-29), so the path shards by batch with NO data-path collective: rank r runs images
shard_range(global_batch, world, r) through its own engine.  The only exchange is ONE all-gather per batch
(RCCL over xGMI on GPUs; gloo in the CPU tests) of the padded detections packed into one buffer of
fixed-size records (include/yolo355.h, "multi-GPU exchange"):
    record = i32 count (-1: padding record of a ragged shard), i32 total (detections the image had: > count when the
             record was cut at max_det; -1 in a padding record), i32 pad[2], f32 boxes[max_det][4],
             f32 scores[max_det], i32 cls[max_det], rounded up to 16 bytes
Every rank sends ceil(global_batch / world) records; result order = global image index.  Calibration happens
once (rank 0) and the 11 exponents (44 bytes) are broadcast -- never per rank on different data.

Two transports with the same wire format: torch.distributed (`allgather_detections`, any backend) and the C ABI
(`RcclGather`: y355_pack_dets / y355_allgather_dets / y355_unpack_dets straight on RCCL, no torch collective).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


def shard_range(global_batch, world_size, rank):
    """contiguous shard of the global batch owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(global_batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def records_per_rank(global_batch, world_size):
    return -(-global_batch // world_size)


def record_bytes(max_det):
    return 16 + (24 * max_det + 15) // 16 * 16


def broadcast_exponents(sa, src=0, device=None):
    """share rank `src`'s 11 activation exponents with every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [int(v) for v in sa]
    t = torch.tensor([int(v) for v in (sa if sa is not None else [0] * 11)], dtype=torch.int32, device=device)
    dist.broadcast(t, src)
    return [int(v) for v in t.cpu()]


def pack_detections(boxes, scores, cls, count, records=None, out=None):
    """padded outputs of a forward (boxes f32 [n,md,4], scores f32 [n,md], cls i32 [n,md], count i32 [n]) ->
    uint8 [records, record_bytes(md)]; entries at or past count[i] are zeroed (equal detections = equal bytes),
    records past n carry count -1.  Works on CPU and GPU tensors (a handful of torch ops; the C ABI's
    y355_pack_dets does the same in one launch)."""
    n, md = scores.shape[0], scores.shape[1]
    records = n if records is None else int(records)
    rb = record_bytes(md)
    rec = out if out is not None else torch.empty((records, rb), dtype=torch.uint8, device=scores.device)
    rec.zero_()
    r32 = rec.view(torch.int32)                                   # [records, rb / 4]
    r32[:, 0] = -1
    r32[:, 1] = -1
    if n:
        cnt = count[:n].to(torch.int32).clamp(0, md)
        r32[:n, 1] = count[:n].to(torch.int32).clamp(min=0)
        keep = torch.arange(md, device=scores.device)[None, :] < cnt[:, None]          # [n, md]
        r32[:n, 0] = cnt
        r32[:n, 4:4 + 4 * md] = (boxes[:n].contiguous().view(torch.int32).reshape(n, md, 4) * keep[:, :, None]).reshape(n, 4 * md)
        r32[:n, 4 + 4 * md:4 + 5 * md] = scores[:n].contiguous().view(torch.int32) * keep
        r32[:n, 4 + 5 * md:4 + 6 * md] = cls[:n].to(torch.int32) * keep
    return rec


def pack_detections_kernel(boxes, scores, cls, count, records, out, max_det=None):
    """the same packing in ONE launch on the caller's current HIP stream (C ABI y355_pack_dets_capped): no torch ops, no stream
    but the current one -- what a per-step gather beside running engines wants (bench.py).  `max_det` below the arrays' own
    cap ships the first max_det detections of every image (anchor order) in records of record_bytes(max_det)."""
    from . import _ffi
    n, md = scores.shape[0], scores.shape[1]
    cap = md if max_det is None else min(int(max_det), md)
    st = C.c_void_p(torch.cuda.current_stream(scores.device).cuda_stream)
    _ffi.check(_ffi.lib().y355_pack_dets_capped(boxes.data_ptr(), scores.data_ptr(), cls.data_ptr(), count.data_ptr(), n, int(records),
                                                md, cap, out.data_ptr(), st))
    return out


def unpack_records(rec, max_det, global_batch=None):
    """uint8 [R, record_bytes] -> (boxes [G,md,4], scores [G,md], cls [G,md], count [G]) dropping padding records
    (count -1); rank-major records of contiguous shards are already in global image order."""
    md = max_det
    r32 = rec.view(torch.int32)
    cnt = r32[:, 0]
    keep = (cnt >= 0).nonzero(as_tuple=True)[0]
    if global_batch is not None and keep.numel() != global_batch:
        raise RuntimeError("gathered %d image records, expected %d" % (keep.numel(), global_batch))
    r32 = r32[keep]
    G = r32.shape[0]
    boxes = r32[:, 4:4 + 4 * md].contiguous().view(torch.float32).reshape(G, md, 4)
    scores = r32[:, 4 + 4 * md:4 + 5 * md].contiguous().view(torch.float32)
    cls = r32[:, 4 + 5 * md:4 + 6 * md].contiguous()
    return boxes, scores, cls, r32[:, 0].contiguous()


def truncated_images(rec):
    """number of image records of a gathered buffer that were cut at the record's cap (header word 1 > word 0)."""
    r32 = rec.view(torch.int32)
    return int(((r32[:, 0] >= 0) & (r32[:, 1] > r32[:, 0])).sum().item())


def allgather_detections(boxes, scores, cls, count, global_batch=None, async_op=False, send=None, recv=None):
    """ONE collective per batch: pack this rank's padded detections (its shard of `global_batch`, default world x n),
    all_gather_into_tensor, and return the gathered padded tensors in global image order.  async_op: returns
    (finish, [work]) where finish() -> the tensors, to be called after work.wait()."""
    world = dist.get_world_size()
    n, md = scores.shape[0], scores.shape[1]
    gb = world * n if global_batch is None else int(global_batch)
    rpr = records_per_rank(gb, world)
    snd = pack_detections(boxes, scores, cls, count, rpr, out=send)
    rcv = recv if recv is not None else torch.empty((world * rpr, snd.shape[1]), dtype=torch.uint8, device=snd.device)
    w = dist.all_gather_into_tensor(rcv, snd, async_op=async_op)

    def finish():
        return unpack_records(rcv, md, gb)
    return (finish, [w]) if async_op else finish()


def unpack(boxes, scores, cls, count):
    """padded tensors -> list of (bboxes [n,4] f32, scores [n] f32, cls_inds [n] i64) numpy."""
    b, s, c, n = (t.cpu().numpy() for t in (boxes, scores, cls, count))
    return [(b[i, :n[i]].copy(), s[i, :n[i]].copy(), c[i, :n[i]].astype(np.int64)) for i in range(len(n))]


class RcclGather:
    """The same exchange through the C ABI (no torch collective): y355_comm_* / y355_pack_dets / y355_allgather_dets /
    y355_unpack_dets on the caller's current HIP stream.  `exchange_id(id_bytes_or_None) -> bytes` ships rank 0's 128-byte
    id to every rank (default: torch.distributed.broadcast_object_list when a process group exists; world 1 needs none)."""

    def __init__(self, world, rank, device, exchange_id=None):
        from . import _ffi
        self._lib = _ffi.lib()
        self._check = _ffi.check
        self.world, self.rank = int(world), int(rank)
        self.device = torch.device(device)
        idbuf = (C.c_char * 128)()
        if self.rank == 0:
            self._check(self._lib.y355_comm_unique_id(idbuf))
        raw = bytes(idbuf)
        if self.world > 1:
            if exchange_id is not None:
                raw = exchange_id(raw if self.rank == 0 else None)
            else:
                box = [raw if self.rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                raw = box[0]
        h = C.c_void_p()
        self._check(self._lib.y355_comm_init(C.byref(h), self.world, self.rank, raw, self.device.index or 0))
        self._h = h

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._lib.y355_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def allgather(self, boxes, scores, cls, count, global_batch=None):
        """padded device tensors of this rank -> gathered padded device tensors in global image order (asynchronous on
        the current stream)."""
        n, md = scores.shape[0], scores.shape[1]
        gb = self.world * n if global_batch is None else int(global_batch)
        rpr = records_per_rank(gb, self.world)
        rb = self._lib.y355_packed_det_bytes(md)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        snd = torch.empty((rpr, rb), dtype=torch.uint8, device=self.device)
        rcv = torch.empty((self.world * rpr, rb), dtype=torch.uint8, device=self.device)
        self._check(self._lib.y355_pack_dets(boxes.data_ptr(), scores.data_ptr(), cls.data_ptr(), count.data_ptr(), n, rpr, md,
                                             snd.data_ptr(), st))
        self._check(self._lib.y355_allgather_dets(self._h, snd.data_ptr(), rcv.data_ptr(), rpr, md, st))
        # records of rank r: its shard's images first, padding after; slot = global image index or -1
        slot = np.full((self.world * rpr,), -1, np.int32)
        for r in range(self.world):
            lo, hi = shard_range(gb, self.world, r)
            slot[r * rpr:r * rpr + hi - lo] = np.arange(lo, hi)
        slot_d = torch.from_numpy(slot).to(self.device)
        ob = torch.zeros((gb, md, 4), dtype=torch.float32, device=self.device)
        os_ = torch.zeros((gb, md), dtype=torch.float32, device=self.device)
        oc = torch.zeros((gb, md), dtype=torch.int32, device=self.device)
        on = torch.zeros((gb,), dtype=torch.int32, device=self.device)
        self._check(self._lib.y355_unpack_dets(rcv.data_ptr(), slot_d.data_ptr(), self.world * rpr, md, ob.data_ptr(), os_.data_ptr(),
                                               oc.data_ptr(), on.data_ptr(), st))
        self._keep = (snd, rcv, slot_d)                     # alive until the stream has consumed them
        return ob, os_, oc, on
