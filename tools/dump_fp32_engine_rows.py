#!/usr/bin/env python3
"""GPU side of the fp32 error budget (round-3 review item 2): the fp32 (exact) engine's detections of the cases whose limits are relaxed in
tests/test_gpu_detector.py (l_640, l_1280, enh_s_128x96; s_1280 as the control), at the rows the reference fixtures hold, written to
tests/golden/engine_fp32_rows.npz.  These are outputs of THIS build's engine (not reference data): the CPU test
tests/test_f64_error_budget.py compares them and the reference fixtures with the float64 evaluation of oracle/skyeye_oracle_f64.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skyeye-aerial-object-detection-using-yolo_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch

from cases import DETECTOR_CASES, variant_of
from helpers import build_detector, detector_params, variant_cfg, variant_enhanced
from seeded import seeded_scene
from skyeye import _native

G = os.path.join(ROOT, "tests", "golden")
SAMPLED = np.load(os.path.join(G, "detectors_sampled.npz"))
out = {"library_sources": np.frombuffer(_native.build_info().encode(), dtype=np.uint8)}
for name in ("l_640", "l_1280", "enh_s_128x96", "s_1280"):
    case = [c for c in DETECTOR_CASES if c["name"] == name][0]
    v = variant_of(case)
    m = build_detector(variant_cfg(v), variant_enhanced(v))
    m.load_state_dict({k: torch.from_numpy(np.asarray(a)) for k, a in detector_params(v).items()}, strict=True)
    m = m.eval().set_precision("fp32")
    h, w = case["hw"]
    x = torch.from_numpy(seeded_scene(case["batch"], h, w, case["seed"]).astype(np.float32) / np.float32(255.0)).cuda()
    det, _ = m(x)
    det = det.cpu().numpy()
    if case["store"] == "full":
        out[f"{name}.det"] = det
    else:
        out[f"{name}.det_rows"] = det.reshape(-1, det.shape[-1])[SAMPLED[f"{name}.rows"]]
    print(name, det.shape, flush=True)
    del m
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "engine_fp32_rows.npz"), **out)
print("wrote gpurun_out/engine_fp32_rows.npz")
