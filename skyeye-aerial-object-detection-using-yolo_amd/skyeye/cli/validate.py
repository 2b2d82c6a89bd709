"""Counterpart of reference skyeye/cli/validate.py (which cannot be imported: SyntaxError at :337, undefined imports
:21-28).  Reproduces its call contract around the hot path -- uint8 batch -> /255 -> model -> non_max_suppression --
and its three timing buckets (validate.py:229-256,323-326).  Dataset / mAP accounting is out of scope:
frames are synthetic unless an ``.npy`` array of uint8 [N,3,H,W] frames is given.

    python -m skyeye.cli.validate --cfg skyeye_s.yaml --img-size 640 --batch-size 32
"""
import argparse

import numpy as np
import torch

from ..core.models import SkyEyeDetector
from ..utils.general import check_img_size
from ..utils.metrics import non_max_suppression
from ..utils.torch_utils import select_device, time_sync


@torch.no_grad()
def validate(weights=None, cfg="skyeye_s.yaml", frames=None, batch_size=32, img_size=640, conf_thres=0.001, iou_thres=0.6,
             half=True, device="", num_batches=4, multi_label=True, single_cls=False, nms_mode="literal", verbose=True, augment=False):
    device = select_device(device)                                           # validate.py:177
    model = SkyEyeDetector(cfg)
    if weights:
        model.load_from_pretrained(weights)                                  # validate.py:184 (load_model)
    img_size = check_img_size(img_size, s=int(model.stride.max()))           # validate.py:188
    if half:
        model.half()                                                         # validate.py:195-197
    model.eval()                                                             # validate.py:200
    if frames is None:
        rng = np.random.default_rng(0)
        frames = rng.integers(0, 256, size=(batch_size * num_batches, 3, img_size, img_size), dtype=np.uint8)
    elif isinstance(frames, str):
        frames = np.load(frames)
    model(torch.zeros(1, 3, img_size, img_size, device=device))              # warm-up, validate.py:209
    dt, seen, kept = [0.0, 0.0, 0.0], 0, 0
    results = []
    for i in range(0, len(frames), batch_size):
        t1 = time_sync()
        img = torch.from_numpy(frames[i:i + batch_size]).to(device, non_blocking=True)   # uint8; /255 happens in the engine (:236-238)
        t2 = time_sync()
        dt[0] += t2 - t1
        out, _train_out = model(img, augment=augment)                        # validate.py:245
        t3 = time_sync()
        dt[1] += t3 - t2
        out = non_max_suppression(out, conf_thres, iou_thres, multi_label=multi_label, agnostic=single_cls, mode=nms_mode)   # :255
        dt[2] += time_sync() - t3
        seen += img.shape[0]
        kept += sum(int(o.shape[0]) for o in out)
        results.extend(out)
    t = tuple(x / max(seen, 1) * 1e3 for x in dt)
    shape = (batch_size, 3, img_size, img_size)
    if verbose:
        print(f"Speed: %.1fms pre-process, %.1fms inference, %.1fms NMS per image at shape {shape}" % t)   # validate.py:323-326
    return dict(speed_ms=t, images=seen, boxes=kept, results=results)


def parse_opt():
    p = argparse.ArgumentParser()
    p.add_argument("--weights", type=str, default=None)
    p.add_argument("--cfg", type=str, default="skyeye_s.yaml")
    p.add_argument("--frames", type=str, default=None, help=".npy of uint8 [N,3,H,W] frames (default: synthetic)")
    p.add_argument("--batch-size", type=int, default=32)
    p.add_argument("--img-size", "--imgsz", "--img", type=int, default=640)
    p.add_argument("--conf-thres", type=float, default=0.001)
    p.add_argument("--iou-thres", type=float, default=0.6)
    p.add_argument("--device", default="")
    p.add_argument("--no-half", action="store_true")
    p.add_argument("--augment", action="store_true", help="augmented inference (validate.py:123)")
    return p.parse_args()


if __name__ == "__main__":
    o = parse_opt()
    validate(o.weights, o.cfg, o.frames, o.batch_size, o.img_size, o.conf_thres, o.iou_thres, not o.no_half, o.device, augment=o.augment)
