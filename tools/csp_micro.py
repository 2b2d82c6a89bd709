#!/usr/bin/env python3
"""Per-launch timing of one CSPBlock(2h, 2h, n) at a detector size (default: the 64-channel bottlenecks at 160 x 160, B = 32)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skyeye-aerial-object-detection-using-yolo_amd")):
    sys.path.insert(0, p)
import torch
from skyeye import _native as N
from skyeye.core.models import CSPBlock
c = int(sys.argv[1]) if len(sys.argv) > 1 else 128
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 160
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
m = CSPBlock(c, c, num_blocks=3).eval().set_precision(prec)
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
x = torch.randn(B, c, hw, hw, device="cuda")
m(x)
h = m._engine([x])
outs = [torch.empty(sh, dtype=torch.float32, device="cuda") for sh in h.output_shapes()]
prof = h.profile_forward([N.buffer_from_tensor(x)], [N.buffer_from_tensor(t) for t in outs], torch.cuda.current_stream().cuda_stream, iters=5)
tag = f"dbg={os.environ.get('SKY_CONV_DBG', '0')} nofuse={os.environ.get('SKY_NO_FUSE_CV1', '0')}"
for i, (ms, fl, t) in enumerate(prof):
    info = h.op_info(i)
    if info.startswith("conv"):
        print(f"{tag} {ms * 1e3:8.1f} us {fl / max(ms, 1e-9) / 1e9:7.1f} TF/s  {info}")
