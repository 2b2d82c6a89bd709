"""Where the fp8 engine's error comes from: relative L2 error against the fp32 engine of single blocks, CSP chains, the backbone stages and the neck (run on the GPU box).
r02: one convolution 3.7 % (bf16 0.23 %), CSP with 9 bottlenecks 8.5 % (0.54 %), backbone P3 / P4 / P5 18 / 24 / 34 % (1.3 / 1.7 / 2.5 %): uniformly 14-16x bf16, i.e. the four mantissa bits."""
import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/skyeye-aerial-object-detection-using-yolo_amd'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import numpy as np, torch
import skyeye.core.models as M
from helpers import load_seeded, build_detector, detector_params, variant_cfg
from seeded import seeded_scene, seeded_input
def rel(a,b): return float((a-b).norm()/b.norm())
x = torch.from_numpy(seeded_input("d.x",(2,128,40,40),3,-2.0,2.0)).cuda()
for name, mk in [("conv3x3 128", lambda: M.ConvolutionBlock(128,128,3,1)), ("conv1x1 128", lambda: M.ConvolutionBlock(128,128,1,1)),
                 ("bottleneck 128", lambda: M.BottleneckBlock(128,128,True,1.0)), ("csp 128 n3", lambda: M.CSPBlock(128,128,3)), ("csp 128 n9", lambda: M.CSPBlock(128,128,9))]:
    ref = load_seeded(mk(), 7).set_precision("fp32")(x)
    for prec in ("bf16","fp8"):
        y = load_seeded(mk(), 7).set_precision(prec)(x)
        print(f"{name:16s} {prec}: rel L2 err {rel(y,ref):.4f}")
# detector stages
P = detector_params("skyeye_s")
frames = torch.from_numpy(seeded_scene(2,320,320,21)).cuda()
outs={}
for prec in ("fp32","bf16","fp8"):
    m = build_detector(variant_cfg("skyeye_s")); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k,v in P.items()}, strict=True); m.eval().set_precision(prec)
    bb = m.backbone.backbone if hasattr(m.backbone,'backbone') else m.backbone
    bb.set_precision(prec)
    f = bb._run([frames])
    outs[prec]=f
    if prec!="fp32":
        print(prec, "backbone P3/P4/P5 rel err:", [round(rel(a,b),4) for a,b in zip(f, outs["fp32"])])
    nk = m.neck; nk.set_precision(prec)
    o = nk._run([t for t in outs["fp32"]])
    outs[prec+"_neck"]=o
    if prec!="fp32": print(prec, "neck alone (fp32 inputs) rel err:", [round(rel(a,b),4) for a,b in zip(o, outs["fp32_neck"])])
    d,_ = m(frames)
    outs[prec+"_det"]=d
    if prec!="fp32":
        r = outs["fp32_det"]
        print(prec, "det logits? obj abs err mean", float((d[...,4]-r[...,4]).abs().mean()))
