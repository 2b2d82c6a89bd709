#!/usr/bin/env python3
"""bf16 engine against the fp32 engine on seeded scenes, continuous error measures (relative L2 of the raw logits per level, mean
objectness error, mean row IoU on confident rows) + the threshold rates the tests assert.  One line of JSON per (model, size).
Run once per library build (SKYEYE_HIP_LIB=.../libskyeye_hip_c<c>.so): tools/pre_scale_ab.sh builds them."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skyeye-aerial-object-detection-using-yolo_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch

from helpers import build_detector, detector_params, variant_cfg
from parity import row_iou
from seeded import seeded_scene


def det(variant, prec):
    m = build_detector(variant_cfg(variant))
    m.load_state_dict({k: torch.from_numpy(np.asarray(a)) for k, a in detector_params(variant).items()}, strict=True)
    return m.eval().set_precision(prec)


tag = os.environ.get("SKYEYE_HIP_LIB", "default").split("libskyeye_hip")[-1]
for variant, hw, B in [("skyeye_s", (640, 640), 2), ("skyeye_s", (1280, 1280), 1), ("skyeye_l", (640, 640), 2), ("skyeye_s", (320, 320), 8)]:
    x = torch.from_numpy(seeded_scene(B, hw[0], hw[1], 21)).cuda()
    ref, rawr = det(variant, "fp32")(x)
    out, rawb = det(variant, "bf16")(x)
    l2 = [float((b - r).norm() / r.norm()) for b, r in zip(rawb, rawr)]
    conf = ref[..., 4] > 0.05
    ri = row_iou(out.cpu().numpy().reshape(-1, out.shape[-1]), ref.cpu().numpy().reshape(-1, ref.shape[-1]), conf.cpu().numpy().reshape(-1))
    print(json.dumps(dict(lib=tag, model=variant, hw=hw, rel_l2_logits=[round(v, 5) for v in l2], mean_dobj=round(float((out[..., 4] - ref[..., 4]).abs().mean()), 6),
                          cls_agree=round(float((out[..., 5:].argmax(-1) == ref[..., 5:].argmax(-1))[conf].float().mean()), 4), rows=int(conf.sum()),
                          row_iou_mean=round(float(ri.mean()), 4), row_iou_gt90=round(float((ri > 0.9).mean()), 4))), flush=True)
