"""N > 1 data path on CPU: world_size-2 gloo processes exercise sharding and the fused box all-gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from skyeye.distributed import all_gather_detections, pack_detections, shard_bounds, unpack_detections


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_result(rank, b_local, max_det):
    g = torch.Generator().manual_seed(1000 + rank)
    rows = torch.rand((b_local, max_det, 7), generator=g)
    counts = torch.randint(0, max_det + 1, (b_local,), generator=g, dtype=torch.int32)
    return rows, counts


def _worker(rank, world, port, b_local, max_det, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, counts = _fake_result(rank, b_local, max_det)
    all_rows, all_counts = all_gather_detections(rows, counts)
    q.put((rank, all_rows.numpy(), all_counts.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_pack_roundtrip_is_bit_exact():
    rows, counts = _fake_result(3, 5, 11)
    r2, c2 = unpack_detections(pack_detections(rows, counts), 11)
    assert torch.equal(rows, r2) and torch.equal(counts, c2)


def test_shards_cover_the_batch_once():
    for n, w in [(256, 8), (10, 4), (3, 8), (32, 1)]:
        seen = []
        for r in range(w):
            lo, hi = shard_bounds(n, r, w)
            seen += list(range(lo, hi))
        assert seen == list(range(n))


@pytest.mark.timeout(120)
def test_all_gather_of_boxes_world2_gloo():
    world, b_local, max_det = 2, 4, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, b_local, max_det, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect_rows = np.concatenate([_fake_result(r, b_local, max_det)[0].numpy() for r in range(world)], 0)
    expect_counts = np.concatenate([_fake_result(r, b_local, max_det)[1].numpy() for r in range(world)], 0)
    for rank, rows, counts in got:
        # every rank holds the same rank-ordered result, bit for bit == the single-process result
        assert np.array_equal(rows.view(np.uint32), expect_rows.view(np.uint32))
        assert np.array_equal(counts, expect_counts)
