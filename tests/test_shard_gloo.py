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
        (gb, gs, gc, gn), works = shard.allgather_detections(boxes, scores, cls, count, async_op=True)
        for w in works:
            w.wait()
        dets = shard.unpack(gb, gs, gc, gn)
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


@pytest.mark.timeout(300)
def test_allgather_of_padded_detections_world2():
    world, gbatch, max_det = 2, 8, 16
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
