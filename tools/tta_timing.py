#!/usr/bin/env python3
"""Timing of the f4 front end on one MI355X: plain forward vs model(x, augment=True) (3 passes) on 32 frames @1280, and
detect_tiled on one 3000 x 4000 aerial frame (12 windows of 1280).  Prints one JSON line per measurement."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from bench import build_model, calibrate_objectness  # noqa: E402
from seeded import seeded_scene  # noqa: E402
from skyeye.utils import tta as T  # noqa: E402
from skyeye.utils.metrics import non_max_suppression  # noqa: E402
from skyeye.utils.torch_utils import scale_img  # noqa: E402


def timed(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


def main():
    dev = torch.device("cuda", 0)
    model, _ = build_model("skyeye_s", "bf16", dev)
    B = int(os.environ.get("TTA_BATCH", 32))
    x = torch.from_numpy(seeded_scene(B, 1280, 1280, seed=3)).to(dev)
    calibrate_objectness(model, x[:4], 0.01, 0.25)
    plain = timed(lambda: model(x))
    aug = timed(lambda: model(x, augment=True))
    s1 = timed(lambda: scale_img(x, 0.83, gs=32, flip=3))
    s2 = timed(lambda: scale_img(x, 0.67, gs=32))
    y = model(x, augment=True)[0]
    nms = timed(lambda: non_max_suppression(y, 0.25, 0.45, mode="corrected"), iters=5)
    print(json.dumps(dict(what="tta", batch=B, plain_ms=round(plain, 3), augment_ms=round(aug, 3), scale_083_flip_ms=round(s1, 3),
                          scale_067_ms=round(s2, 3), rows=int(y.shape[1]), nms_ms=round(nms, 3),
                          frames_per_s_augment=round(B / aug * 1e3, 1))))
    frame = torch.from_numpy(np.ascontiguousarray(seeded_scene(1, 3000, 4000, seed=4)[0].transpose(1, 2, 0))).to(dev)
    org = T.tile_origins(3000, 4000, 1280, 1280, 0.2)
    raw = timed(lambda: T.detect_tiled(model, frame, tile=1280, overlap=0.2, return_raw=True), iters=10)
    full = timed(lambda: T.detect_tiled(model, frame, tile=1280, overlap=0.2, conf_thres=0.25), iters=10)
    rows = T.detect_tiled(model, frame, tile=1280, overlap=0.2, conf_thres=0.25)
    og = torch.from_numpy(org).to(dev)
    gather = timed(lambda: T.tile_gather(frame, og, 1280, 1280))
    print(json.dumps(dict(what="tiled", frame=[3000, 4000], tiles=len(org), gather_ms=round(gather, 3), forward_and_map_ms=round(raw, 3),
                          with_nms_ms=round(full, 3), boxes=int(rows.shape[0]), megapixels_per_s=round(12.0 / full * 1e3, 1))))


if __name__ == "__main__":
    main()
