#!/usr/bin/env python3
"""What the NMS costs the pipelined loop: the two-slice forward pass alone (captured, four passes per replay) against bench.py's default loop in the same process.
usage (GPU box): python experiments/forward_only_probe.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skyeye-aerial-object-detection-using-yolo_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch

from bench import build_model, calibrate_objectness
from skyeye.utils.torch_utils import capture_graph

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
model, _ = build_model("skyeye_s", "bf16", dev)
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, size=(32, 3, 1280, 1280), dtype=np.uint8)).to(dev)
calibrate_objectness(model, x, 0.01, 0.25)
model.reuse_output_buffers(True)
model.parallel_slices(2)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


fwd, _ = capture_graph(lambda: tuple(model(x, return_raw=False)[0] for _ in range(4)), warmup=2)
pipe, _ = capture_graph(lambda: tuple(model.detect_nms_pipelined(x, 0.25, 0.45, max_detections=300, parity=i & 1) for i in range(4)), warmup=2)
strict, _ = capture_graph(lambda: model.detect_nms(x, 0.25, 0.45, max_detections=300), warmup=2)
for r in range(reps):
    a = timed(fwd.replay, 15) / 4
    b = timed(pipe.replay, 15) / 4
    c = timed(strict.replay, 60)
    print(f"rep {r}: forward only {32 / a:8.1f} frames/s ({a * 1e3:.3f} ms)   pipelined forward + NMS {32 / b:8.1f} ({b * 1e3:.3f} ms)   strict order {32 / c:8.1f} ({c * 1e3:.3f} ms)", flush=True)
