#!/usr/bin/env python3
"""Single-convolution timing harness: ConvolutionBlock through the engine, hipEvent time of the conv launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skyeye-aerial-object-detection-using-yolo_amd")):
    sys.path.insert(0, p)
import torch
from skyeye import _native as N
from skyeye.core.models import ConvolutionBlock

cases = [(128, 128, 1, 1, 160), (128, 128, 3, 1, 80), (64, 64, 1, 1, 320), (64, 64, 3, 1, 160), (256, 256, 3, 1, 40), (32, 32, 3, 1, 320), (16, 32, 3, 1, 640), (64, 128, 3, 2, 320), (128, 256, 3, 2, 160), (128, 128, 3, 2, 160), (256, 512, 3, 2, 80), (512, 512, 1, 1, 40), (1024, 512, 1, 1, 40), (512, 256, 1, 1, 80), (256, 256, 1, 1, 80), (768, 512, 1, 1, 40), (384, 256, 1, 1, 80), (256, 128, 1, 1, 160), (256, 256, 1, 1, 40), (128, 128, 1, 1, 80), (64, 64, 1, 1, 160), (128, 384, 1, 1, 160), (256, 768, 1, 1, 80), (128, 512, 1, 1, 160), (256, 512, 1, 1, 80)]
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
if len(sys.argv) > 2:
    cases = [cases[int(i)] for i in sys.argv[2].split(",")]
iters = int(os.environ.get("MICRO_ITERS", "5"))
B = int(os.environ.get("MICRO_B", "32"))
for cin, cout, k, s, hw in cases:
    m = ConvolutionBlock(cin, cout, k, s, activation=os.environ.get("MICRO_NOACT") is None).eval().set_precision(prec)
    x = torch.randn(B, cin, hw, hw, device="cuda")
    m(x)
    h = m._engine([x])
    outs = [torch.empty(sh, dtype=torch.float32, device="cuda") for sh in h.output_shapes()]
    prof = h.profile_forward([N.buffer_from_tensor(x)], [N.buffer_from_tensor(t) for t in outs], torch.cuda.current_stream().cuda_stream, iters=iters)
    ms, fl, _ = prof[1]
    byts = B * hw * hw * (cin + cout) * (2 if prec == "bf16" else 4)
    print(f"B={B} dbg={os.environ.get('SKY_CONV_DBG','0'):>2} {prec} conv{k}x{k} {cin}->{cout} @{hw}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s  {byts/ms/1e9:7.2f} TB/s(in+out)")
