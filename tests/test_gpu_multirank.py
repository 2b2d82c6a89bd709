"""Multi-GPU readiness on ONE MI355X: the code that the first 8-GPU run executes -- RCCL
initialisation, the fused all-gather of box buffers, bench.py's step -- runs here on a world-size-1 `nccl` process group, and
the tiled path is checked as "the tiles of a frame on two virtual ranks == one rank, exactly"."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from helpers import build_detector, detector_params, variant_cfg

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(precision="bf16"):
    P = detector_params("skyeye_s")
    m = build_detector(variant_cfg("skyeye_s"))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()}, strict=True)
    return m.eval().set_precision(precision)


def _frame(h, w, seed):
    from seeded import seeded_scene
    f = seeded_scene(1, h, w, seed)[0]                   # [3, h, w] uint8
    return torch.from_numpy(np.ascontiguousarray(f.transpose(1, 2, 0))).cuda()


def test_rows_mode_nms_matches_oracle_bit_exact():
    """sky_nms mode 2 (the cross-tile stage: rows are already boxes) against the oracle's restatement."""
    sys.path.insert(0, ROOT)
    from oracle import skyeye_oracle as O
    from skyeye.utils.metrics import non_max_suppression
    g = np.random.default_rng(5)
    n = 4000
    xy = g.uniform(0, 2000, (n, 2)).astype(np.float32)
    wh = g.uniform(4, 200, (n, 2)).astype(np.float32)
    rows = np.zeros((1, n, 7), np.float32)
    rows[0, :, 0:2] = xy
    rows[0, :, 2:4] = xy + wh
    rows[0, :, 4] = g.uniform(0, 1, n).astype(np.float32)
    rows[0, :, 5] = g.integers(0, 10, n).astype(np.float32)
    rows[0, ::7, 4] = 0.0                                 # empty slots, as past the count of a tile block
    rows[0, 5:60:5, 4] = rows[0, 5, 4]                    # ties
    got = non_max_suppression(torch.from_numpy(rows).cuda(), 0.25, 0.45, max_det=1000, mode="rows")[0].cpu().numpy()
    ref = O.non_max_suppression(rows, 0.25, 0.45, max_detections=1000, mode="rows")[0]
    assert got.shape == ref.shape and got.shape[0] > 50
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_tiled_two_virtual_ranks_equal_one_rank_exactly():
    """"1-GPU tiled == 8-GPU tiled, exactly".  The tile set of one frame is run as two virtual ranks (each its
    contiguous shard, own batch size), the blocks are laid out as the all-gather lays them out, the owner merges: the rows must
    equal detect_tiled_sharded on one rank bit for bit."""
    from skyeye.utils.tta import detect_tiled_sharded, merge_tile_survivors, tile_origins, tile_shard, tile_survivors
    m = _model("bf16")
    frame = _frame(608, 800, 3)
    tile, ov, conf = 256, 0.25, 0.05
    one = detect_tiled_sharded(m, frame, tile=tile, overlap=ov, conf_thres=conf, tile_max_det=100)
    org = tile_origins(608, 800, tile, tile, ov)
    assert len(org) >= 9
    for world in (2, 3):
        blocks_r, blocks_c = [], []
        for rank in range(world):
            lo, hi, per = tile_shard(len(org), rank, world)
            r, c = tile_survivors(m, frame, org[lo:hi], (tile, tile), per, conf_thres=conf, tile_max_det=100)
            blocks_r.append(r)
            blocks_c.append(c)
        merged = merge_tile_survivors(torch.cat(blocks_r, 0), torch.cat(blocks_c, 0), conf_thres=conf)
        assert merged.shape == one.shape and one.shape[0] > 0
        assert torch.equal(merged.view(torch.int32), one.view(torch.int32)), f"world {world} differs from one rank"
    # boxes are in frame pixels: some survivor must come from a tile that does not start at the origin
    assert float(one[:, 2].max()) > tile


def test_rccl_world1_all_gather_and_bench_step():
    """RCCL init + all_gather_into_tensor + pack / unpack on the real backend (world size 1): the first 8-GPU run is then not
    the first time this code executes on this stack."""
    import torch.distributed as dist
    from skyeye.distributed import all_gather_detections, pack_detections, unpack_detections
    from skyeye.utils.metrics import nms_raw
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        assert int(one.item()) == 1
        m = _model("bf16")
        from seeded import seeded_scene
        x = torch.from_numpy(seeded_scene(2, 256, 256, 11)).cuda()
        det, _ = m(x, return_raw=False)
        rows, counts = nms_raw(det, 0.05, 0.45, max_detections=300)
        # the fused buffer through the collective itself (all_gather_detections short-cuts world == 1)
        packed = pack_detections(rows, counts)
        out = torch.empty((1,) + tuple(packed.shape), dtype=packed.dtype, device=dev)
        dist.all_gather_into_tensor(out, packed)
        r2, c2 = unpack_detections(out.reshape(packed.shape[0], -1), rows.shape[1], rows.shape[2])
        assert torch.equal(r2.view(torch.int32), rows.view(torch.int32)) and torch.equal(c2, counts)
        r3, c3 = all_gather_detections(rows, counts)
        assert torch.equal(r3, rows) and torch.equal(c3, counts)
        # BoxExchange: the NMS kernel writes the exchanged block in place (image-strided outputs through the C ABI), the block goes
        # through the collective itself (always_collective); the captured form runs in a child process (next test)
        from skyeye.distributed import BoxExchange
        ex = BoxExchange(2, 300, 7, dev, always_collective=True)
        assert ex.collective and ex.gathered.data_ptr() != ex.local.data_ptr()
        nms_raw(det, 0.05, 0.45, max_detections=300, out=ex.rows, counts=ex.counts)
        ar, ac = ex.gather()
        assert torch.equal(ar.contiguous().view(torch.int32), rows.view(torch.int32)) and torch.equal(ac.contiguous(), counts)
        dist.barrier()
    finally:
        dist.destroy_process_group()


_GRAPH_CHILD = r"""
import os, socket, sys
ROOT = sys.argv[1]
for p in (ROOT, os.path.join(ROOT, "skyeye-aerial-object-detection-using-yolo_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
import torch.distributed as dist
from helpers import build_detector, detector_params, variant_cfg
from seeded import seeded_scene
from skyeye.distributed import BoxExchange
from skyeye.utils.metrics import nms_raw
from skyeye.utils.torch_utils import capture_graph

with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
m = build_detector(variant_cfg("skyeye_s"))
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in detector_params("skyeye_s").items()}, strict=True)
m = m.eval().set_precision("bf16").reuse_output_buffers(True).parallel_slices(2)
x = torch.from_numpy(seeded_scene(2, 256, 256, 11)).cuda()
det, _ = m(x, return_raw=False)
rows, counts = nms_raw(det.clone(), 0.05, 0.45, max_detections=300)
ex = BoxExchange(2, 300, 7, dev, always_collective=True)

def step():
    m.detect_nms(x, 0.05, 0.45, max_detections=300, out=(ex.rows, ex.counts))
    return ex.gather()

graph, (gr, gc) = capture_graph(step, warmup=2)
ex.local.zero_(); ex.gathered.zero_()
graph.replay()
torch.cuda.synchronize()
assert int(counts.sum()) > 0
assert torch.equal(gr.contiguous().view(torch.int32), rows.view(torch.int32)) and torch.equal(gc.contiguous(), counts)
print("forward + NMS + RCCL all-gather replayed from one hipGraph: equal to the eager result")
del graph, gr, gc
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("done")
"""


def test_rccl_all_gather_inside_a_captured_graph_world1():
    """forward (two slices) + NMS writing the exchange block in place + the RCCL all-gather of that block, captured as ONE hipGraph and
    replayed (world size 1, always_collective).  In a child process: tearing a communicator down around a graph that captured one of
    its collectives aborted the interpreter once when it ran inside the suite's process."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", _GRAPH_CHILD, ROOT], capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout[-1500:])
    sys.stderr.write(r.stderr[-1500:])
    assert "replayed from one hipGraph" in r.stdout, f"child exited with {r.returncode}"


def test_return_raw_false_skips_raw_levels_and_keeps_detections():
    m = _model("bf16")
    from seeded import seeded_scene
    x = torch.from_numpy(seeded_scene(2, 128, 160, 2)).cuda()
    det_a, raw = m(x)
    det_b, none = m(x, return_raw=False)
    assert len(raw) == 3 and none == []
    assert torch.equal(det_a.view(torch.int32), det_b.view(torch.int32))


def test_default_build_ignores_conv_dbg_env():
    """The shipped library has no SKY_CONV_DBG code path (kernel experiments need -DSKY_EXPERIMENTS): the variable changes nothing."""
    import skyeye.core.models as M
    from helpers import load_seeded
    from seeded import seeded_input
    m = load_seeded(M.ConvolutionBlock(128, 128, 3, 1), 5).set_precision("bf16")
    x = torch.from_numpy(seeded_input("dbg.x", (2, 128, 48, 48), 5, -2.0, 2.0)).cuda()
    ref = m(x).cpu()
    for v in ("1", "15", "2"):                           # former stage switches: skip MFMA taps / halo DMA / epilogue
        os.environ["SKY_CONV_DBG"] = v
        try:
            m.refresh_weights()                          # new plan under the variable
            got = m(x).cpu()
        finally:
            os.environ.pop("SKY_CONV_DBG", None)
        assert torch.equal(got, ref)


def test_graph_replay_equals_eager():
    from skyeye.utils.metrics import nms_raw
    from skyeye.utils.torch_utils import capture_graph
    from seeded import seeded_scene
    m = _model("bf16")
    x = torch.from_numpy(seeded_scene(2, 256, 256, 4)).cuda()

    def step():
        det, _ = m(x, return_raw=False)
        return nms_raw(det, 0.05, 0.45)
    rows_e, counts_e = step()
    rows_e, counts_e = rows_e.clone(), counts_e.clone()
    graph, (rows_g, counts_g) = capture_graph(step)
    rows_g.zero_(); counts_g.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert int(counts_e.sum()) > 0
    assert torch.equal(counts_g, counts_e) and torch.equal(rows_g.view(torch.int32), rows_e.view(torch.int32))
