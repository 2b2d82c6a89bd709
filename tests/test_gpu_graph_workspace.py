"""A captured hipGraph replays launches that point INTO the NMS workspace of the utility handle it was captured with.  Round 2 fixed a
use-after-free there (b876bd9: sky_nms freed an outgrown workspace with hipFree while a captured graph could still replay into it;
now the old block is retired with the handle) and left one graph-replay fault of a dropped experiment unexplained (DESIGN.md
section 3, "Graph replay and engine-owned memory").  This test drives exactly that path, once: capture forward-free NMS on a small
batch, make the SAME handle's workspace grow (a larger geometry through the C ABI), overwrite the new block by running the larger
problem, replay the graph: results equal the eager results of the small batch, bit for bit, and the device is healthy afterwards."""
import ctypes

import numpy as np
import pytest
import torch

from skyeye import _native as N
from skyeye.utils.metrics import nms_raw
from skyeye.utils.torch_utils import capture_graph

pytestmark = pytest.mark.gpu


def _det(B, n, seed):
    rng = np.random.default_rng(seed)
    d = np.zeros((B, n, 15), np.float32)
    d[..., 0:2] = rng.uniform(50, 600, (B, n, 2))
    d[..., 2:4] = rng.uniform(10, 80, (B, n, 2))
    d[..., 4] = rng.uniform(0, 1, (B, n)) ** 4
    d[..., 5:] = rng.uniform(0, 1, (B, n, 10))
    return torch.from_numpy(d).cuda()


def test_graph_replay_survives_nms_workspace_growth():
    small = _det(2, 3000, 1)
    eager_rows, eager_counts = nms_raw(small, 0.25, 0.45)
    eager_rows, eager_counts = eager_rows.clone(), eager_counts.clone()
    graph, (rows, counts) = capture_graph(lambda: nms_raw(small, 0.25, 0.45), warmup=2)
    keep = graph._sky_keep
    assert len(keep) >= 1, "capture_graph must keep the utility handle of the captured NMS alive"
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(counts, eager_counts) and torch.equal(rows, eager_rows)
    # grow the workspace of the handle the graph was captured with: 8 x 60 000 rows need a far larger block than 2 x 3 000
    h = keep[0]
    big = _det(8, 60000, 2)
    p = N.SkyNmsParams()
    p.struct_size = ctypes.sizeof(N.SkyNmsParams)
    p.conf_threshold, p.iou_threshold, p.max_detections, p.max_nms, p.max_wh, p.mode = 0.25, 0.45, 300, 30000, 4096.0, 0
    out = torch.empty((8, 300, 7), dtype=torch.float32, device="cuda")
    cnt = torch.empty((8,), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(2):        # the second call runs entirely in the new block
        N.check(h.L.sky_nms(h.h, big.data_ptr(), 8, 60000, 10, ctypes.byref(p), out.data_ptr(), cnt.data_ptr(), ctypes.c_void_p(stream)), h.h)
    torch.cuda.synchronize()
    assert int(cnt.min()) > 0
    # the graph still points into the retired block: it must be alive and give the same answer
    rows.zero_()
    counts.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(counts, eager_counts) and torch.equal(rows, eager_rows)
    # and the handle's own eager path (new block) agrees as well
    r2, c2 = nms_raw(small, 0.25, 0.45)
    assert torch.equal(c2, eager_counts) and torch.equal(r2, eager_rows)
