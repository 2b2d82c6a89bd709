"""non_max_suppression -- drop-in for reference skyeye/utils/metrics.py:361-457, computed on the MI355X.

``mode='literal'`` (default) reproduces the file as written: boxes stay (cx, cy, w, h) and are suppressed as if they
were corners, the score is objectness alone, the per-class offset uses the class *confidence* column and rows are
7 wide for nc > 1.  ``mode='corrected'`` gives the YOLOv5 semantics the code imitates:
conf = obj * cls, corner boxes, class-id offset, rows [x1, y1, x2, y2, conf, cls].
"""
import ctypes

import torch

from .. import _native as N

import collections

_UTIL = collections.OrderedDict()       # (device, stream pointer) -> utility Handle, least recently used first
_UTIL_MAX = 8
_KEEP = []                              # stack of lists: capture_graph collects the handles a captured region used


def _handle(device_index, stream=0):
    """One utility handle (it owns the NMS workspace, tens of MB) per (device, stream): two streams never share a workspace.

    The table is a small LRU: a stream pointer can be recycled by a NEW stream after the old one is destroyed, and every
    ``capture_graph`` call makes a fresh stream, so entries must not pile up.  An evicted handle is only dropped here; it is
    closed when its last owner lets go of it.  A captured hipGraph replays into the workspace of the handle it was captured
    with: ``capture_graph`` therefore keeps that handle alive with the graph (``graph._sky_keep``), whatever this table does."""
    key = (device_index, stream)
    h = _UTIL.get(key)
    if h is None:
        h = N.Handle(N.make_config("UTILITY", device=device_index))
        _UTIL[key] = h
        while len(_UTIL) > _UTIL_MAX:
            _UTIL.popitem(last=False)
    else:
        _UTIL.move_to_end(key)
    if _KEEP:
        _KEEP[-1].append(h)
    return h


def forget_stream(device_index, stream):
    """Drop the utility handle of a stream that is going away (its pointer may be handed to a new stream)."""
    _UTIL.pop((device_index, stream), None)


def nms_raw(prediction, conf_threshold=0.25, iou_threshold=0.45, classes=None, agnostic=False, multi_label=False,
            max_detections=300, mode="literal", max_nms=30000, max_wh=4096.0, out=None, counts=None):
    """Asynchronous form: returns (rows [B, max_det, 7] float32, counts [B] int32) on the device, no host sync.  ``out`` / ``counts``:
    tensors (or batch slices of them) to write into instead of fresh ones; the image dimension may be strided (the views of a
    ``skyeye.distributed.BoxExchange`` block: the kernel writes straight into the buffer the all-gather sends)."""
    if not prediction.is_cuda:
        raise N.SkyEyeNativeError("non_max_suppression: prediction must be on the HIP device (no CPU path)")
    pred = prediction.float().contiguous()
    B, Nrows, no = pred.shape
    p = N.SkyNmsParams()
    p.struct_size = ctypes.sizeof(N.SkyNmsParams)
    p.conf_threshold, p.iou_threshold = float(conf_threshold), float(iou_threshold)
    p.agnostic, p.multi_label = int(bool(agnostic)), int(bool(multi_label))
    p.max_detections, p.max_nms, p.max_wh = int(max_detections), int(max_nms), float(max_wh)
    p.mode = {"literal": 0, "corrected": 1, "rows": 2}[mode]
    cls = [] if classes is None else [int(c) for c in classes]
    p.n_classes = len(cls)
    for i, c in enumerate(cls):
        p.classes[i] = c
    # the kernel defines every element (rows past the count are zeroed, k_nms.hip: nms_greedy_kernel): no fill launches here
    if out is None:
        out = torch.empty((B, max_detections, 7), dtype=torch.float32, device=pred.device)
    if counts is None:
        counts = torch.empty((B,), dtype=torch.int32, device=pred.device)
    if tuple(out.shape) != (B, max_detections, 7) or out.dtype != torch.float32 or out.stride()[1:] != (7, 1) or \
            (B > 1 and out.stride(0) < max_detections * 7) or tuple(counts.shape) != (B,) or counts.dtype != torch.int32 or \
            (B > 1 and counts.stride(0) < 1):
        raise N.SkyEyeNativeError("nms_raw: out must be float32 [B, max_detections, 7] with dense rows and counts int32 [B] "
                                  "(only the image dimension may be strided)")
    p.out_image_stride = int(out.stride(0)) if B > 1 else 0
    p.counts_stride = int(counts.stride(0)) if B > 1 else 0
    stream = torch.cuda.current_stream(pred.device).cuda_stream
    h = _handle(pred.device.index or 0, stream)
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


# --------------------------------------------------------------------------------------------------------------------
# Evaluation accounting: box_iou, process_batch, compute_ap, ap_per_class.
# The pairwise IoU runs on the MI355X (sky_box_iou); the AP bookkeeping is host-side numpy in the reference
# (metrics.py:124-225) and stays host-side here.
def box_iou(box1, box2, layout="literal"):
    """Pairwise IoU -> [N, M] (reference metrics.py:17-44), computed on the device.

    ``layout='literal'``: box1 is indexed the way the file indexes it, ``box1[0..3]`` = x1, y1, x2, y2 *rows*, i.e. a
    [4, N] tensor, although its docstring says (N, 4).  ``layout='rows'``: box1 is [N, 4].
    box2 is [M, 4] in both.  Heights carry the file's + 1e-7, the union one more."""
    if not (box1.is_cuda and box2.is_cuda):
        raise N.SkyEyeNativeError("box_iou: boxes must be on the HIP device (no CPU path)")
    a = box1.float().contiguous()
    b = box2.float().contiguous()
    four_by_n = layout == "literal"
    if a.dim() != 2 or b.dim() != 2 or b.shape[1] != 4 or a.shape[0 if four_by_n else 1] != 4:
        raise ValueError(f"box_iou: box1 {tuple(a.shape)} / box2 {tuple(b.shape)} do not fit layout '{layout}'")
    n, m = a.shape[1 if four_by_n else 0], b.shape[0]
    out = torch.empty((n, m), dtype=torch.float32, device=a.device)
    h = _handle(a.device.index or 0)
    stream = torch.cuda.current_stream(a.device).cuda_stream
    N.check(h.L.sky_box_iou(h.h, a.data_ptr(), n, int(four_by_n), b.data_ptr(), m, out.data_ptr(), ctypes.c_void_p(stream)), h.h)
    return out


def process_batch(detections, labels, iouv):
    """Correct-prediction matrix [n, len(iouv)] (reference validate.py:71-108).

    detections [n, 6] = (x1, y1, x2, y2, conf, cls), labels [m, 5] = (cls, x1, y1, x2, y2), iouv = IoU thresholds.
    The reference body cannot run (box_iou is handed [m, 4] where it indexes [4, m]; ``torch.unique`` has no
    ``return_index``): this is the YOLOv5 rule it imitates -- per threshold, candidate (label, detection) pairs of the
    same class with IoU >= threshold, best IoU first, each detection and each label used once.  IoU on the device, the
    (small) matching on the host.  PARITY UNPINNED against the reference (no runnable form); pinned against oracle/."""
    import numpy as np
    iouv_h = iouv.detach().cpu().numpy() if torch.is_tensor(iouv) else np.asarray(iouv)
    n, m = detections.shape[0], labels.shape[0]
    correct = np.zeros((n, iouv_h.shape[0]), dtype=bool)
    if n and m:
        iou = box_iou(labels[:, 1:5], detections[:, :4], layout="rows").cpu().numpy()            # [m, n]
        same = labels[:, 0:1].cpu().numpy() == detections[:, 5].cpu().numpy()[None, :]
        for i, thr in enumerate(iouv_h):
            li, di = np.nonzero((iou >= thr) & same)
            if li.size:
                matches = np.stack([li, di, iou[li, di]], 1)
                if li.size > 1:
                    matches = matches[np.argsort(-matches[:, 2], kind="stable")]
                    matches = matches[np.unique(matches[:, 1], return_index=True)[1]]
                    matches = matches[np.unique(matches[:, 0], return_index=True)[1]]
                correct[matches[:, 1].astype(np.int64), i] = True
    dev = iouv.device if torch.is_tensor(iouv) else detections.device
    return torch.from_numpy(correct).to(dev)


def compute_ap(recall, precision):
    """(AP, precision envelope, recall) of one precision/recall curve (reference metrics.py:124-148): sentinels
    (0, 0) and (1, 0), running maximum from the right, area where recall changes."""
    import numpy as np
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([0.0], precision, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    i = np.where(mrec[1:] != mrec[:-1])[0]
    ap = np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])
    return ap, mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, plot=False, save_dir=None, names=()):
    """Per-class precision, recall, AP [classes, len(iouv)], F1 and the class list (reference metrics.py:151-225).
    Detections are ranked by confidence; P/R curves are sampled on 1000 confidence points from the IoU-0.5 column; the
    returned P, R, F1 are taken where the class-mean F1 peaks.  ``plot`` is accepted and ignored (plots are out of scope)."""
    import numpy as np
    tp, conf, pred_cls, target_cls = np.asarray(tp), np.asarray(conf), np.asarray(pred_cls), np.asarray(target_cls)
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    unique_classes = np.unique(target_cls)
    nc = unique_classes.shape[0]
    px = np.linspace(0, 1, 1000)
    ap = np.zeros((nc, tp.shape[1]))
    precision = np.zeros((nc, 1000))
    recall = np.zeros((nc, 1000))
    for ci, c in enumerate(unique_classes):
        sel = pred_cls == c
        n_gt = (target_cls == c).sum()
        if sel.sum() == 0 or n_gt == 0:
            continue
        fpc = (1 - tp[sel]).cumsum(0)
        tpc = tp[sel].cumsum(0)
        recall_curve = tpc / (n_gt + 1e-16)
        recall[ci] = np.interp(-px, -conf[sel], recall_curve[:, 0])
        precision_curve = tpc / (tpc + fpc)
        precision[ci] = np.interp(-px, -conf[sel], precision_curve[:, 0])
        for j in range(tp.shape[1]):
            ap[ci, j], _, _ = compute_ap(recall_curve[:, j], precision_curve[:, j])
    f1 = 2 * precision * recall / (precision + recall + 1e-16)
    i = f1.mean(0).argmax()
    return precision[:, i], recall[:, i], ap, f1[:, i], unique_classes


def mean_average_precision(predictions, labels, iouv=None):
    """validate.py:262-318 for one list of images: predictions = per-image [n, 6] (x1, y1, x2, y2, conf, cls) device
    tensors, labels = per-image [m, 5] (cls, x1, y1, x2, y2).  -> dict(mp, mr, map50, map, ap [classes, 10], classes)."""
    import numpy as np
    dev = predictions[0].device if len(predictions) else torch.device("cuda")
    if iouv is None:
        iouv = torch.linspace(0.5, 0.95, 10, device=dev)                          # validate.py:203
    stats = []
    for pred, lab in zip(predictions, labels):
        tcls = lab[:, 0].cpu().numpy() if lab.shape[0] else np.zeros((0,))
        if pred.shape[0] == 0:
            if lab.shape[0]:
                stats.append((np.zeros((0, iouv.shape[0]), dtype=bool), np.zeros((0,)), np.zeros((0,)), tcls))
            continue
        correct = process_batch(pred, lab, iouv) if lab.shape[0] else torch.zeros((pred.shape[0], iouv.shape[0]), dtype=torch.bool)
        stats.append((correct.cpu().numpy(), pred[:, 4].cpu().numpy(), pred[:, 5].cpu().numpy(), tcls))
    if not stats:
        return dict(mp=0.0, mr=0.0, map50=0.0, map=0.0, ap=np.zeros((0, iouv.shape[0])), classes=np.zeros((0,)))
    tp, conf, pcls, tcls = [np.concatenate(x, 0) for x in zip(*stats)]             # validate.py:305
    if not tp.shape[0] or not tp.any():
        return dict(mp=0.0, mr=0.0, map50=0.0, map=0.0, ap=np.zeros((0, iouv.shape[0])), classes=np.unique(tcls))
    p, r, ap, f1, classes = ap_per_class(tp, conf, pcls, tcls)
    return dict(mp=float(p.mean()), mr=float(r.mean()), map50=float(ap[:, 0].mean()), map=float(ap.mean()), ap=ap, classes=classes)
