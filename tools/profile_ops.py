#!/usr/bin/env python3
"""Per-launch table of the planned graph (hipEvent timing inside libskyeye_hip.so): ms, TFLOP/s, description."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skyeye-aerial-object-detection-using-yolo_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch

from bench import build_model
from skyeye import _native as N

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="skyeye_s")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=1280)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
model, _ = build_model(a.model, a.precision, dev)
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, size=(a.batch, 3, a.size, a.size), dtype=np.uint8)).to(dev)
model(x)
h = model._engine([x])
outs = [torch.empty(s, dtype=torch.float32, device=dev) for s in h.output_shapes()]
stream = torch.cuda.current_stream(dev).cuda_stream
prof = h.profile_forward([N.buffer_from_tensor(x)], [N.buffer_from_tensor(t) for t in outs], stream, iters=a.iters)
tot = sum(p[0] for p in prof)
print(f"# {a.model} {a.precision} B={a.batch} @{a.size}: {len(prof)} launches, {tot:.3f} ms per forward, stats {h.stats()}")
for i, (ms, fl, tag) in enumerate(prof):
    print(f"{i:3d} {ms:8.4f} ms {100 * ms / tot:5.1f}% {fl / (ms * 1e-3) / 1e12 if ms > 0 else 0:8.1f} TF/s  {h.op_info(i)}")
