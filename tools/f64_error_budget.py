#!/usr/bin/env python3
"""fp32 error budget of the exact-mode exceptions (round-3 review item 2).  For a detector case: the float64 evaluation of the graph
(oracle/skyeye_oracle_f64.py, CPU) against (a) the REFERENCE's fixture (PyTorch-CPU fp32: oneDNN convolutions) and (b) the fp32 engine's
output of one GPU run (tests/golden/engine_fp32_rows.npz, tools/dump_fp32_engine_rows.py), in the units of tests/parity.det_close:
worst |d| / (1e-4 x column scale) and 1 - min IoU on boxes of at least 8 px.  The three pairwise distances (engine - fixture, f64 -
fixture, f64 - engine) say whose rounding a margin above 1 is.

    python tools/f64_error_budget.py [case ...]        (default: enh_s_128x96 l_640; l_1280 takes ~2 min and ~8 GB)
    -> profiles/r04_f64_error_budget.json is what tests/test_f64_error_budget.py and the limits in tests/test_gpu_detector.py cite"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skyeye-aerial-object-detection-using-yolo_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np

from cases import DETECTOR_CASES, variant_of
from helpers import detector_params
from parity import level_scales
from seeded import seeded_scene
from oracle import skyeye_oracle_f64 as O64

G = os.path.join(ROOT, "tests", "golden")


def distance(a, b, scales, tol=1e-4):
    """det_close's two measures between detections a and b (b supplies the scales): worst |d| / limit, 1 - min IoU (boxes >= 8 px)."""
    a, b = a.astype(np.float64), b.astype(np.float64)
    d = np.abs(a - b)
    lim = np.full(b.shape, tol)
    lim[..., 0:2] = tol * np.maximum(np.abs(b[..., 0:2]), scales[..., 0:1])
    lim[..., 2:4] = tol * np.maximum(np.abs(b[..., 2:4]), scales[..., 1:3])

    def corners(t):
        return t[..., 0] - t[..., 2] / 2, t[..., 1] - t[..., 3] / 2, t[..., 0] + t[..., 2] / 2, t[..., 1] + t[..., 3] / 2
    ax1, ay1, ax2, ay2 = corners(a)
    bx1, by1, bx2, by2 = corners(b)
    inter = np.clip(np.minimum(ax2, bx2) - np.maximum(ax1, bx1), 0, None) * np.clip(np.minimum(ay2, by2) - np.maximum(ay1, by1), 0, None)
    union = (ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter
    sized = (b[..., 2] >= 8.0) & (b[..., 3] >= 8.0)
    iou = np.where(sized, inter / np.maximum(union, 1e-30), 1.0)
    return {"worst_ratio_at_1e-4": round(float((d / lim).max()), 4), "one_minus_min_iou": float(1.0 - iou.min()),
            "class_equal": bool(np.array_equal(a[..., 5:].argmax(-1), b[..., 5:].argmax(-1)))}


def budget(name, engine=None):
    case = [c for c in DETECTOR_CASES if c["name"] == name][0]
    P = detector_params(variant_of(case))
    h, w = case["hw"]
    frames = seeded_scene(case["batch"], h, w, case["seed"])
    t0 = time.time()
    det64, _ = O64.detector_forward(P, frames.astype(np.float64) / 255.0, 10, enhanced=bool(case.get("enhanced")))
    secs = time.time() - t0
    scales = level_scales(case["hw"])
    if case["store"] == "full":
        ref = np.load(os.path.join(G, "detectors_full.npz"))[f"{name}.det"]
        f64 = det64
        sc = scales
        eng = None if engine is None or f"{name}.det" not in engine else engine[f"{name}.det"]
    else:
        S = np.load(os.path.join(G, "detectors_sampled.npz"))
        rows = S[f"{name}.rows"]
        ref = S[f"{name}.det_rows"]
        f64 = det64.reshape(-1, det64.shape[-1])[rows]
        sc = np.tile(scales, (case["batch"], 1))[rows]
        eng = None if engine is None or f"{name}.det_rows" not in engine else engine[f"{name}.det_rows"]
    # the true division x / 255 of the reference rounds the frame to fp32 once; the f64 graph keeps the exact quotient: part of (a)
    out = {"case": name, "rows": int(ref.shape[0] if ref.ndim == 2 else np.prod(ref.shape[:-1])), "f64_seconds": round(secs, 1),
           "f64_vs_reference_fixture": distance(f64, ref, sc)}
    if eng is not None:
        out["engine_vs_reference_fixture"] = distance(eng, ref, sc)
        out["f64_vs_engine"] = distance(f64, eng, sc)
    return out


if __name__ == "__main__":
    names = sys.argv[1:] or ["enh_s_128x96", "l_640"]
    ef = os.path.join(G, "engine_fp32_rows.npz")
    engine = np.load(ef) if os.path.exists(ef) else None
    res = [budget(n, engine) for n in names]
    for r in res:
        print(json.dumps(r))
