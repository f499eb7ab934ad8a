"""Multi-process test of the batch-shard path on CPU: world_size 2, gloo.
The per-rank detections are synthetic padded tensors -- the collective, the shard arithmetic
and the result order (global image index) are what is under test; the GPU kernels are not."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yolo355 import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_dets(global_idx, max_det):
    """deterministic padded detections of one image"""
    rng = np.random.default_rng(1000 + global_idx)
    n = int(rng.integers(0, max_det + 1))
    b = np.zeros((max_det, 4), np.float32)
    s = np.zeros((max_det,), np.float32)
    c = np.zeros((max_det,), np.int32)
    b[:n] = rng.random((n, 4), dtype=np.float32)
    s[:n] = rng.random(n, dtype=np.float32)
    c[:n] = rng.integers(0, 20, n)
    return b, s, c, n


def _worker(rank, world, port, global_batch, max_det, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard.shard_range(global_batch, world, rank)
        per = [_fake_dets(i, max_det) for i in range(lo, hi)]
        boxes = torch.from_numpy(np.stack([p[0] for p in per]))
        scores = torch.from_numpy(np.stack([p[1] for p in per]))
        cls = torch.from_numpy(np.stack([p[2] for p in per]))
        count = torch.tensor([p[3] for p in per], dtype=torch.int32)
        sa = shard.broadcast_exponents([5, 5, 6, 7, 8, 8, 10, 10, 11, 11, 4] if rank == 0 else None, 0)
        # ONE collective per batch: the four tensors travel packed in one buffer of fixed-size records
        calls = []
        orig = dist.all_gather_into_tensor

        def counting(*a, **k):
            calls.append(1)
            return orig(*a, **k)
        dist.all_gather_into_tensor = counting
        finish, works = shard.allgather_detections(boxes, scores, cls, count, global_batch=global_batch, async_op=True)
        dist.all_gather_into_tensor = orig
        for w in works:
            w.wait()
        gb, gs, gc, gn = finish()
        dets = shard.unpack(gb, gs, gc, gn)
        assert len(calls) == 1 and len(works) == 1
        ok = sa == [5, 5, 6, 7, 8, 8, 10, 10, 11, 11, 4] and len(dets) == global_batch
        for i in range(global_batch):
            b, s, c, n = _fake_dets(i, max_det)
            ok = ok and np.array_equal(dets[i][0], b[:n]) and np.array_equal(dets[i][1], s[:n]) \
                and np.array_equal(dets[i][2], c[:n].astype(np.int64))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_the_batch():
    for gb, w in [(512, 8), (64, 1), (10, 4), (3, 8)]:
        r = [shard.shard_range(gb, w, k) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == gb
        assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
        assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= 1
    assert shard.shard_range(512, 8, 3) == (192, 256)        # rank r gets images [64r, 64r+64)


def test_pack_unpack_roundtrip_and_record_layout():
    """the wire format of include/yolo355.h ("multi-GPU exchange"): 16-byte header (count, the image's own count, 0, 0) + boxes +
    scores + cls, rounded up to 16 bytes; entries past count zeroed; padding records carry -1 / -1 and are dropped"""
    for md in (16, 7):
        per = [_fake_dets(i, md) for i in range(3)]
        boxes = torch.from_numpy(np.stack([p[0] for p in per]))
        scores = torch.from_numpy(np.stack([p[1] for p in per]))
        cls = torch.from_numpy(np.stack([p[2] for p in per]))
        count = torch.tensor([p[3] for p in per], dtype=torch.int32)
        scores[0, per[0][3]:] = 9.0                          # garbage past count must not travel
        rec = shard.pack_detections(boxes, scores, cls, count, records=5)
        assert rec.shape == (5, shard.record_bytes(md)) and shard.record_bytes(md) % 16 == 0
        assert shard.record_bytes(16) == 16 + 24 * 16 and shard.record_bytes(7) == 16 + 176
        r32 = rec.view(torch.int32)
        assert r32[:, 0].tolist() == [per[0][3], per[1][3], per[2][3], -1, -1] and int(r32[:, 2:4].abs().sum()) == 0
        assert r32[:, 1].tolist() == r32[:, 0].tolist() and shard.truncated_images(rec) == 0      # nothing was cut
        big = count.clone()
        big[1] = md + 5                                       # an image with more detections than the record holds
        assert shard.truncated_images(shard.pack_detections(boxes, scores, cls, big, records=5)) == 1
        b, s, c, n = shard.unpack_records(rec, md, 3)
        assert n.tolist() == [p[3] for p in per]
        for i, p in enumerate(per):
            assert np.array_equal(b[i].numpy()[:p[3]], p[0][:p[3]]) and np.array_equal(s[i].numpy()[:p[3]], p[1][:p[3]])
            assert np.array_equal(c[i].numpy()[:p[3]], p[2][:p[3]]) and float(s[i, p[3]:].abs().sum()) == 0.0


@pytest.mark.timeout(300)
@pytest.mark.parametrize("gbatch", [8, 7])
def test_allgather_of_padded_detections_world2(gbatch):
    """world 2, gloo; gbatch 7 = ragged shards (4 + 3 images: the short rank sends a padding record)"""
    world, max_det = 2, 16
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, gbatch, max_det, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
