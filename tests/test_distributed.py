"""N > 1 data path on CPU: gloo process groups of 2 and 8 ranks exercise sharding, the fused box all-gather (the legacy packed form and
BoxExchange, whose buffers are allocated once and which the NMS kernel writes in place) and the tile-survivor gather of tiled inference."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from skyeye.distributed import BoxExchange, all_gather_detections, pack_detections, shard_bounds, unpack_detections


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


# ---- tiled inference: the tiles of one frame on different ranks (SURVEY 8e "Tiled (C5)") -------------------------------------
def _fake_survivors(n_tiles, R):
    """deterministic per-tile survivor blocks (what stage 1 of skyeye.utils.tta.detect_tiled_sharded produces)"""
    g = torch.Generator().manual_seed(77)
    counts = torch.randint(0, R + 1, (n_tiles,), generator=g, dtype=torch.int32)
    rows = torch.rand((n_tiles, R, 7), generator=g)
    for t in range(n_tiles):
        rows[t, counts[t]:] = 0.0
    return rows, counts


def _tile_worker(rank, world, port, n_tiles, R, q):
    from skyeye.utils.tta import tile_shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    all_rows, all_counts = _fake_survivors(n_tiles, R)
    lo, hi, per = tile_shard(n_tiles, rank, world)
    rows = torch.zeros((per, R, 7))
    counts = torch.zeros((per,), dtype=torch.int32)
    rows[: hi - lo] = all_rows[lo:hi]
    counts[: hi - lo] = all_counts[lo:hi]
    g_rows, g_counts = all_gather_detections(rows, counts)
    q.put((rank, g_rows.numpy(), g_counts.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_shards_cover_the_tiles_in_order():
    from skyeye.utils.tta import tile_shard
    for n, w in [(12, 8), (12, 2), (5, 4), (1, 8), (9, 1)]:
        seen, pers = [], set()
        for r in range(w):
            lo, hi, per = tile_shard(n, r, w)
            assert hi - lo <= per
            seen += list(range(lo, hi))
            pers.add(per)
        assert seen == list(range(n)) and len(pers) == 1


@pytest.mark.timeout(120)
def test_tile_survivor_gather_world2_gloo_keeps_tile_order():
    """After the gather every rank holds the survivor blocks of ALL tiles in row-major tile order (empty blocks where a rank
    had fewer tiles than the common block count): dropping the empty blocks gives exactly the 1-process sequence."""
    world, n_tiles, R = 2, 5, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, world, port, n_tiles, R, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows1, counts1 = _fake_survivors(n_tiles, R)
    from skyeye.utils.tta import tile_shard
    per = tile_shard(n_tiles, 0, world)[2]
    real = []                                            # positions of the real blocks inside the gathered buffer
    for r in range(world):
        lo, hi, _ = tile_shard(n_tiles, r, world)
        real += [r * per + k for k in range(hi - lo)]
    for rank, rows, counts in got:
        assert rows.shape == (world * per, R, 7)
        assert np.array_equal(rows[real].view(np.uint32), rows1.numpy().view(np.uint32))
        assert np.array_equal(counts[real], counts1.numpy())
        empty = sorted(set(range(world * per)) - set(real))
        assert not counts[empty].any() and not rows[empty].any()


# ---- world size 8: the geometry of BASELINE.json configs[3] (B = 256 -> 32 frames per rank) through BoxExchange ----------------
def _exchange_worker(rank, world, port, n_items, max_det, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(n_items, rank, world)
    rows_all, counts_all = _fake_result(7, n_items, max_det)          # what ONE process would hold for the whole batch
    ex = BoxExchange(hi - lo, max_det, 7)
    ptr = ex.local.data_ptr()
    for step in range(2):                                              # two steps through the SAME buffers (nothing is allocated per step)
        ex.rows.copy_(rows_all[lo:hi] + step)                          # the NMS kernel writes these views in place on the GPU
        ex.counts.copy_(counts_all[lo:hi])
        all_rows, all_counts = ex.gather()
        assert ex.local.data_ptr() == ptr and all_rows.data_ptr() == ex.gathered.data_ptr()
    q.put((rank, all_rows.clone().numpy(), all_counts.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_box_exchange_world8_gloo_b256_equals_one_process():
    world, n_items, max_det = 8, 256, 20
    assert [shard_bounds(n_items, r, world) for r in (0, 7)] == [(0, 32), (224, 256)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, n_items, max_det, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows1, counts1 = _fake_result(7, n_items, max_det)
    rows1 = (rows1 + 1).numpy()
    for rank, rows, counts in got:
        assert rows.shape == (n_items, max_det, 7)
        assert np.array_equal(rows.view(np.uint32), rows1.view(np.uint32)) and np.array_equal(counts, counts1.numpy())


def test_box_exchange_single_process_views():
    """No process group: the gathered views ARE the local block; rows / counts are strided views of ONE int32 buffer."""
    ex = BoxExchange(3, 4, 7)
    ex.rows.copy_(torch.arange(3 * 4 * 7, dtype=torch.float32).reshape(3, 4, 7))
    ex.counts.copy_(torch.tensor([1, 0, 4], dtype=torch.int32))
    assert ex.rows.stride() == (4 * 7 + 1, 7, 1) and ex.counts.stride() == (4 * 7 + 1,)
    assert ex.local[1, :4].view(torch.float32).tolist() == [28.0, 29.0, 30.0, 31.0] and ex.local[:, -1].tolist() == [1, 0, 4]
    r, c = ex.gather()
    assert r.data_ptr() == ex.rows.data_ptr() and torch.equal(c, ex.counts)


@pytest.mark.timeout(300)
def test_tile_survivor_gather_world8_12_tiles_uneven_shards():
    """12 tiles (a 3000 x 4000 frame) over 8 ranks: four ranks hold two tiles, four hold one and an empty block; dropping the empty
    blocks of the gathered buffer gives the 1-process tile sequence bit for bit."""
    from skyeye.utils.tta import tile_shard
    world, n_tiles, R = 8, 12, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, world, port, n_tiles, R, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows1, counts1 = _fake_survivors(n_tiles, R)
    per = tile_shard(n_tiles, 0, world)[2]
    sizes = [tile_shard(n_tiles, r, world)[1] - tile_shard(n_tiles, r, world)[0] for r in range(world)]
    assert per == 2 and sorted(sizes) == [1, 1, 1, 1, 2, 2, 2, 2]
    real = []
    for r in range(world):
        lo, hi, _ = tile_shard(n_tiles, r, world)
        real += [r * per + k for k in range(hi - lo)]
    for rank, rows, counts in got:
        assert rows.shape == (world * per, R, 7)
        assert np.array_equal(rows[real].view(np.uint32), rows1.numpy().view(np.uint32)) and np.array_equal(counts[real], counts1.numpy())
        empty = sorted(set(range(world * per)) - set(real))
        assert len(empty) == 4 and not counts[empty].any() and not rows[empty].any()
