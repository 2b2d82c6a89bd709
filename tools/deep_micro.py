#!/usr/bin/env python3
"""Timing of the 256-channel bottleneck 3x3 at skyeye_l's 80 x 80 / B = 32 shape (CSPBlock(512, 512, 3)): deep-pipelined kernel
against the halo-tile kernel (SKY_NO_DEEP3X3=1), per-launch table of both."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skyeye-aerial-object-detection-using-yolo_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch

import skyeye.core.models as M
from helpers import load_seeded
from seeded import seeded_input
from skyeye import _native as N

x = torch.from_numpy(seeded_input("dpm.x", (32, 512, 80, 80), 3, -2.0, 2.0)).cuda()
for deep in (True, False, True, False):
    if not deep:
        os.environ["SKY_NO_DEEP3X3"] = "1"
    m = load_seeded(M.CSPBlock(512, 512, num_blocks=3), 23).set_precision("bf16")
    y = m(x)
    h = m._engine([x])
    os.environ.pop("SKY_NO_DEEP3X3", None)
    outs = [torch.empty(s, dtype=torch.float32, device="cuda") for s in h.output_shapes()]
    prof = h.profile_forward([N.buffer_from_tensor(x)], [N.buffer_from_tensor(t) for t in outs], torch.cuda.current_stream().cuda_stream, iters=20)
    print("deep" if deep else "halo", "total %.4f ms" % sum(p[0] for p in prof))
    for i, (ms, fl, tag) in enumerate(prof):
        if "3x3" in h.op_info(i):
            print(f"   {i:2d} {ms:8.4f} ms {fl / (ms * 1e-3) / 1e12 if ms > 0 else 0:8.1f} TF/s  {h.op_info(i)}")
