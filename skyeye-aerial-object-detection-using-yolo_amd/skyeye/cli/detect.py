"""Counterpart of reference skyeye/cli/detect.py:31-223 (undefined imports at :22-28, needs cv2): the call contract
of ``run`` around the hot path -- frame -> /255 -> model -> non_max_suppression(max_det=) -> scale_boxes -- with the
per-image timing line (detect.py:217-218).  Image decoding / drawing are out of scope; frames come as uint8 arrays."""
import numpy as np
import torch

from ..core.models import SkyEyeDetector
from ..utils.general import check_img_size, scale_boxes, xywh2xyxy
from ..utils.metrics import non_max_suppression
from ..utils.torch_utils import select_device, time_sync


@torch.no_grad()
def run(weights=None, cfg="skyeye_s.yaml", source=None, imgsz=640, conf_thres=0.25, iou_thres=0.45, max_det=1000, device="",
        classes=None, agnostic_nms=False, half=False, orig_shapes=None, nms_mode="corrected", augment=False):
    device = select_device(device)
    model = SkyEyeDetector(cfg)
    if weights:
        model.load_from_pretrained(weights)
    imgsz = check_img_size(imgsz, s=int(model.stride.max()))                  # detect.py:111
    if half:
        model.half()
    model.eval()
    if source is None:
        source = np.random.default_rng(0).integers(0, 256, size=(4, 3, imgsz, imgsz), dtype=np.uint8)
    elif isinstance(source, str):
        source = np.load(source)
    model.warmup(imgsz=(1, 3, imgsz, imgsz))                                  # detect.py:126
    dt, seen, out_all = [0.0, 0.0, 0.0], 0, []
    for i, frame in enumerate(source):
        t1 = time_sync()
        im = torch.from_numpy(np.ascontiguousarray(frame)).to(device)[None]   # uint8 [1,3,H,W]; /255 in the engine (:131-135)
        t2 = time_sync()
        dt[0] += t2 - t1
        pred, _ = model(im, augment=augment, visualize=False)                 # detect.py:140
        t3 = time_sync()
        dt[1] += t3 - t2
        pred = non_max_suppression(pred, conf_thres, iou_thres, classes, agnostic_nms, max_det=max_det, mode=nms_mode)   # :145
        dt[2] += time_sync() - t3
        det = pred[0]
        if nms_mode == "literal" and det.shape[0]:
            det = torch.cat([xywh2xyxy(det[:, :4]), det[:, 4:]], 1)           # the file as written keeps xywh
        if det.shape[0] and orig_shapes is not None:
            det[:, :4] = scale_boxes(im.shape[2:], det[:, :4], orig_shapes[i]).round()    # detect.py:168
        out_all.append(det)
        seen += 1
    t = tuple(x / max(seen, 1) * 1e3 for x in dt)
    print(f"Speed: %.1fms pre-process, %.1fms inference, %.1fms NMS per image at shape {(1, 3, imgsz, imgsz)}" % t)
    return out_all
