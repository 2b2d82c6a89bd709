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
