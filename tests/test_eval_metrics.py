"""Evaluation accounting, CPU side: the host functions of the drop-in package and the oracle against
fixtures the REFERENCE's own functions produced (tests/golden/eval.npz: box_iou metrics.py:17-44, compute_ap :124-148,
ap_per_class :151-225, run in the build container by tests/golden/make_golden.py eval)."""
import os

import numpy as np

from skyeye.utils import metrics as M

E = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval.npz"))


def _oracle():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import skyeye_oracle as O
    return O


def test_compute_ap_matches_reference():
    for k in range(3):
        ap, mpre, mrec = M.compute_ap(E[f"ap{k}.recall"], E[f"ap{k}.precision"])
        assert ap == E[f"ap{k}.ap"]                      # float64 host arithmetic in the same order: exact
        assert np.array_equal(mpre, E[f"ap{k}.mpre"]) and np.array_equal(mrec, E[f"ap{k}.mrec"])


def test_ap_per_class_matches_reference():
    p, r, ap, f1, cls = M.ap_per_class(E["apc.tp"], E["apc.conf"], E["apc.pred_cls"], E["apc.target_cls"])
    assert np.array_equal(cls, E["apc.classes"])
    for got, key in ((p, "apc.p"), (r, "apc.r"), (ap, "apc.ap"), (f1, "apc.f1")):
        assert np.array_equal(got, E[key]), key
    assert ap.shape == (9, 10)                                                # class 0 has no labels


def test_oracle_box_iou_matches_reference():
    O = _oracle()
    out = O.box_iou(E["iou.a"].T, E["iou.b"], literal=True)
    assert np.array_equal(out, E["iou.out"])
    assert np.array_equal(O.box_iou(E["iou.a"], E["iou.b"], literal=False), E["iou.out"])
    assert out[7, 5] > 0.999999 and out[0].max() == 0.0                      # the duplicate box; a zero-area box


def test_oracle_process_batch_hand_case():
    O = _oracle()
    labels = np.array([[1, 0, 0, 10, 10], [1, 20, 20, 30, 30], [2, 0, 0, 10, 10]], np.float32)
    dets = np.array([[0, 0, 10, 10, 0.9, 1],        # exact match of label 0
                     [1, 1, 10, 10, 0.8, 1],        # second-best for label 0: a label is used once
                     [20, 20, 30, 31, 0.7, 1],      # IoU 10/11 with label 1
                     [0, 0, 10, 10, 0.6, 3]], np.float32)   # right box, wrong class
    iouv = np.linspace(0.5, 0.95, 10)
    c = O.process_batch(dets, labels, iouv)
    assert c[0].all() and not c[1].any() and not c[3].any()
    assert c[2, :9].all() and not c[2, 9]            # 0.909 >= 0.90 but < 0.95
    assert O.process_batch(dets[:0], labels, iouv).shape == (0, 10)
