"""non_max_suppression -- drop-in for reference skyeye/utils/metrics.py:361-457, computed on the MI355X.

``mode='literal'`` (default) reproduces the file as written: boxes stay (cx, cy, w, h) and are suppressed as if they
were corners, the score is objectness alone, the per-class offset uses the class *confidence* column and rows are
7 wide for nc > 1.  ``mode='corrected'`` gives the YOLOv5 semantics the code imitates:
conf = obj * cls, corner boxes, class-id offset, rows [x1, y1, x2, y2, conf, cls].
"""
import ctypes

import torch

from .. import _native as N

_UTIL = {}


def _handle(device_index):
    h = _UTIL.get(device_index)
    if h is None:
        h = N.Handle(N.make_config("UTILITY", device=device_index))
        _UTIL[device_index] = h
    return h


def nms_raw(prediction, conf_threshold=0.25, iou_threshold=0.45, classes=None, agnostic=False, multi_label=False,
            max_detections=300, mode="literal", max_nms=30000, max_wh=4096.0):
    """Asynchronous form: returns (rows [B, max_det, 7] float32, counts [B] int32) on the device, no host sync."""
    if not prediction.is_cuda:
        raise N.SkyEyeNativeError("non_max_suppression: prediction must be on the HIP device (no CPU path)")
    pred = prediction.float().contiguous()
    B, Nrows, no = pred.shape
    p = N.SkyNmsParams()
    p.struct_size = ctypes.sizeof(N.SkyNmsParams)
    p.conf_threshold, p.iou_threshold = float(conf_threshold), float(iou_threshold)
    p.agnostic, p.multi_label = int(bool(agnostic)), int(bool(multi_label))
    p.max_detections, p.max_nms, p.max_wh = int(max_detections), int(max_nms), float(max_wh)
    p.mode = 0 if mode == "literal" else 1
    cls = [] if classes is None else [int(c) for c in classes]
    p.n_classes = len(cls)
    for i, c in enumerate(cls):
        p.classes[i] = c
    out = torch.zeros((B, max_detections, 7), dtype=torch.float32, device=pred.device)
    counts = torch.zeros((B,), dtype=torch.int32, device=pred.device)
    h = _handle(pred.device.index or 0)
    stream = torch.cuda.current_stream(pred.device).cuda_stream
    N.check(h.L.sky_nms(h.h, pred.data_ptr(), B, Nrows, no - 5, ctypes.byref(p), out.data_ptr(), counts.data_ptr(),
                        ctypes.c_void_p(stream)), h.h)
    return out, counts


def non_max_suppression(prediction, conf_threshold=0.25, iou_threshold=0.45, classes=None, agnostic=False,
                        multi_label=False, max_detections=300, mode="literal", conf_thres=None, iou_thres=None, max_det=None):
    """-> list of [n, 6|7] tensors, one per image (reference metrics.py:361-369).  ``conf_thres`` / ``iou_thres`` /
    ``max_det`` are the spellings the reference's own callers use (detect.py:145, validate.py:255)."""
    if conf_thres is not None:
        conf_threshold = conf_thres
    if iou_thres is not None:
        iou_threshold = iou_thres
    if max_det is not None:
        max_detections = max_det
    nc = prediction.shape[2] - 5
    out, counts = nms_raw(prediction, conf_threshold, iou_threshold, classes, agnostic, multi_label, max_detections, mode)
    cols = 7 if (mode == "literal" and nc > 1) else 6
    host = counts.cpu().tolist()        # the one host synchronisation of the post-processing step
    res = []
    for b, n in enumerate(host):
        res.append(out[b, :n, :cols] if n else torch.zeros((0, 6), device=prediction.device))
    return res
