"""fp32 error budget of the exact-mode exceptions (round-3 review item 2): for the three detector cases whose IoU limits in
tests/test_gpu_detector.py are above 1e-4 (l_640, l_1280, enh_s_128x96) and one control (s_1280), the float64 evaluation of the graph
(oracle/skyeye_oracle_f64.py) against the REFERENCE's fixture and against the fp32 engine's output of one GPU run
(tests/golden/engine_fp32_rows.npz, written by tools/dump_fp32_engine_rows.py), in det_close's units.  Prints the three pairwise
distances per case and pins them to profiles/r04_f64_error_budget.json, which the limits cite.  CPU only."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from f64_error_budget import budget          # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
COMMITTED = {r["case"]: r for r in json.load(open(os.path.join(ROOT, "profiles", "r04_f64_error_budget.json")))["cases"]}


@pytest.mark.parametrize("name", ["enh_s_128x96", "l_640", "l_1280", "s_1280"])
def test_f64_apportions_the_fp32_difference(name):
    engine = np.load(os.path.join(G, "engine_fp32_rows.npz"))
    r = budget(name, engine)
    ref, eng, pair = r["f64_vs_reference_fixture"], r["f64_vs_engine"], r["engine_vs_reference_fixture"]
    print(f"\n{name}: 1 - min IoU   reference fixture vs f64 {ref['one_minus_min_iou']:.3e}   fp32 engine vs f64 {eng['one_minus_min_iou']:.3e}   "
          f"engine vs fixture {pair['one_minus_min_iou']:.3e}      worst |d| / (1e-4 x scale): {ref['worst_ratio_at_1e-4']:.3f} / "
          f"{eng['worst_ratio_at_1e-4']:.3f} / {pair['worst_ratio_at_1e-4']:.3f}")
    c = COMMITTED[name]
    for k in ("f64_vs_reference_fixture", "f64_vs_engine", "engine_vs_reference_fixture"):
        # the f64 graph is deterministic up to the BLAS's summation order (1e-12 relative): the committed numbers must reproduce
        assert r[k]["one_minus_min_iou"] == pytest.approx(c[k]["one_minus_min_iou"], rel=1e-3, abs=1e-9), k
        assert r[k]["worst_ratio_at_1e-4"] == pytest.approx(c[k]["worst_ratio_at_1e-4"], rel=1e-3, abs=1e-4), k
    # the float64 graph IS the reference's graph: it sits inside the 1e-4 column tolerance of every reference fixture it covers
    assert ref["worst_ratio_at_1e-4"] < 1.0
    # neither fp32 implementation is closer to the other than to the truth by accident: the difference between them is bounded by
    # the sum of their own distances from float64 (the max-statistics are sub-additive up to the row where each maximum sits)
    assert pair["one_minus_min_iou"] <= 1.25 * (ref["one_minus_min_iou"] + eng["one_minus_min_iou"])
