"""GPU parity for non_max_suppression: bit-exact against the reference wrapper's fixtures (tests/golden/nms.npz)
and against the CPU oracle on fresh inputs, literal and corrected modes."""
import os

import numpy as np
import pytest
import torch

from cases import NMS_CASES
from nms_inputs import make_predictions
from oracle import skyeye_oracle as O
from skyeye.utils.metrics import non_max_suppression

pytestmark = pytest.mark.gpu
NMS = np.load(os.path.join(os.path.dirname(__file__), "golden", "nms.npz"))


def pred_for(case):
    return make_predictions(case["nc"], case["batch"], case["n"], case["seed"], ties=case.get("ties", False),
                            distinct_scores=(case["name"] == "over_cap"))


@pytest.mark.parametrize("case", NMS_CASES, ids=[c["name"] for c in NMS_CASES])
def test_nms_matches_reference_wrapper_bit_exact(case):
    pred = pred_for(case)
    res = non_max_suppression(torch.from_numpy(pred).cuda(), **case["kwargs"])
    counts = NMS[f"{case['name']}.counts"]
    assert [int(r.shape[0]) for r in res] == counts.tolist()
    rows = NMS[f"{case['name']}.rows"]
    got = [r.cpu().numpy() for r in res if r.shape[0]]
    if got:
        got = np.concatenate(got, 0)
        assert got.shape == rows.shape
        assert np.array_equal(got.view(np.uint32), rows.view(np.uint32))


@pytest.mark.parametrize("mode", ["literal", "corrected"])
@pytest.mark.parametrize("multi_label", [False, True])
def test_nms_matches_oracle_bit_exact(mode, multi_label):
    pred = make_predictions(7, 3, 5000, 4242)
    kw = dict(conf_threshold=0.2, iou_threshold=0.5, multi_label=multi_label, max_detections=300)
    ref = O.non_max_suppression(pred, mode=mode, **kw)
    res = non_max_suppression(torch.from_numpy(pred).cuda(), mode=mode, **kw)
    for a, b in zip(res, ref):
        assert a.shape[0] == b.shape[0]
        if b.shape[0]:
            assert np.array_equal(a.cpu().numpy().view(np.uint32), b.view(np.uint32))


def test_nms_full_size_sorted_and_idempotent():
    # BASELINE size: 100 800 rows per image (1280x1280), conf 0.001 -> every row is a candidate (30 000 cap)
    pred = make_predictions(10, 2, 100800, 99, distinct_scores=True)
    res = non_max_suppression(torch.from_numpy(pred).cuda(), conf_threshold=0.001, iou_threshold=0.6)
    ref = O.non_max_suppression(pred, conf_threshold=0.001, iou_threshold=0.6)
    for a, b in zip(res, ref):
        a = a.cpu().numpy()
        assert np.all(np.diff(a[:, 4]) <= 0), "kept rows must be in descending score order"
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _crowded(nc, batch, n, seed, clusters=40):
    """Rows whose boxes pile up on a few centres: most candidates are suppressed, so the greedy scan visits thousands of candidates
    (many 512-candidate steps of the workgroup kernel) before it has kept max_detections boxes."""
    r = np.random.default_rng(seed)
    pred = np.zeros((batch, n, nc + 5), np.float32)
    for b in range(batch):
        cx = r.uniform(100, 1100, clusters)
        cy = r.uniform(100, 1100, clusters)
        k = r.integers(0, clusters, n)
        pred[b, :, 0] = cx[k] + r.normal(0, 6, n)
        pred[b, :, 1] = cy[k] + r.normal(0, 6, n)
        pred[b, :, 2] = r.uniform(60, 90, n)
        pred[b, :, 3] = r.uniform(60, 90, n)
        pred[b, :, 4] = r.permutation(n).astype(np.float32) / n * 0.7 + 0.3        # distinct objectness
        pred[b, :, 5:] = r.uniform(0.5, 1.0, (n, nc))
    return pred


@pytest.mark.parametrize("mode", ["literal", "corrected"])
@pytest.mark.parametrize("max_det", [1, 17, 300, 4096])
def test_nms_crowded_scenes_and_kept_list_sizes_match_oracle(mode, max_det):
    pred = _crowded(3, 2, 6000, 11)
    kw = dict(conf_threshold=0.3, iou_threshold=0.45, max_detections=max_det)
    ref = O.non_max_suppression(pred, mode=mode, **kw)
    res = non_max_suppression(torch.from_numpy(pred).cuda(), mode=mode, **kw)
    for a, b in zip(res, ref):
        assert a.shape[0] == b.shape[0] and a.shape[0] <= max_det
        if b.shape[0]:
            assert np.array_equal(a.cpu().numpy().view(np.uint32), b.view(np.uint32))


def test_nms_without_candidates_returns_empty_rows():
    pred = make_predictions(4, 3, 2000, 5)
    res = non_max_suppression(torch.from_numpy(pred).cuda(), conf_threshold=2.0, iou_threshold=0.5)       # nothing passes
    assert [int(r.shape[0]) for r in res] == [0, 0, 0]


def test_nms_strided_outputs_and_the_round3_struct():
    """sky_nms_params ends in out_image_stride / counts_stride (round 4): rows and counts written with an image stride (the blocks of a
    skyeye.distributed.BoxExchange) are the dense result bit for bit, the gaps between the blocks stay untouched; a caller built against the
    round-3 struct (recognised by struct_size) still gets dense outputs; strides below the dense sizes are refused."""
    import ctypes
    from skyeye import _native as N
    from skyeye.distributed import BoxExchange
    from skyeye.utils.metrics import _handle, nms_raw
    pred = torch.from_numpy(make_predictions(5, 4, 3000, 99)).cuda()
    rows, counts = nms_raw(pred, 0.2, 0.45, max_detections=50)
    ex = BoxExchange(4, 50, 7, "cuda")
    ex.local.fill_(-7)
    nms_raw(pred, 0.2, 0.45, max_detections=50, out=ex.rows, counts=ex.counts)
    assert torch.equal(ex.rows.contiguous().view(torch.int32), rows.view(torch.int32)) and torch.equal(ex.counts.contiguous(), counts)
    assert ex.local.shape == (4, 351) and int(counts.sum()) > 0
    # a batch slice of the strided views (what detect_nms hands to a slice's NMS)
    ex.local.fill_(-7)
    nms_raw(pred[2:], 0.2, 0.45, max_detections=50, out=ex.rows[2:], counts=ex.counts[2:])
    assert torch.equal(ex.rows[2:].contiguous().view(torch.int32), rows[2:].view(torch.int32)) and bool((ex.local[:2] == -7).all())
    # the round-3 struct: everything up to `classes`, struct_size says so
    p = N.SkyNmsParams()
    p.struct_size = N.SkyNmsParams.out_image_stride.offset
    p.conf_threshold, p.iou_threshold, p.max_detections, p.max_nms, p.max_wh, p.mode = 0.2, 0.45, 50, 30000, 4096.0, 0
    p.out_image_stride, p.counts_stride = 12345, 77          # (behind the declared size: must be ignored)
    out = torch.empty((4, 50, 7), dtype=torch.float32, device="cuda")
    cnt = torch.empty((4,), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    h = _handle(0, stream)
    N.check(h.L.sky_nms(h.h, pred.data_ptr(), 4, 3000, 5, ctypes.byref(p), out.data_ptr(), cnt.data_ptr(), ctypes.c_void_p(stream)), h.h)
    assert torch.equal(out.view(torch.int32), rows.view(torch.int32)) and torch.equal(cnt, counts)
    p.struct_size = ctypes.sizeof(N.SkyNmsParams)
    p.out_image_stride, p.counts_stride = 50 * 7 - 1, 1
    with pytest.raises(N.SkyEyeNativeError):
        N.check(h.L.sky_nms(h.h, pred.data_ptr(), 4, 3000, 5, ctypes.byref(p), out.data_ptr(), cnt.data_ptr(), ctypes.c_void_p(stream)), h.h)
