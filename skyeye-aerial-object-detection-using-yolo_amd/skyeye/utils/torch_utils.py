"""time_sync / select_device / scale_img of reference skyeye/utils/torch_utils.py (:70-118 the timing convention of the CLIs,
:262-288 the resize of test-time augmentation)."""
import ctypes
import math
import time

import torch

from .. import _native as N


def time_sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()


def select_device(device=""):
    if str(device).lower() == "cpu" or not torch.cuda.is_available():
        raise RuntimeError("the SkyEye HIP engine needs an MI355X: there is no CPU path (reference select_device, torch_utils.py:70-106)")
    idx = 0 if device in ("", "cuda") else int(str(device).replace("cuda:", "").split(",")[0])
    return torch.device("cuda", idx)


def scale_img_geometry(h, w, ratio, same_shape=False, gs=32):
    """-> (resized (h, w), padded (h, w)): torch_utils.py:275-288 (python float arithmetic, ``int()`` truncation, ``math.ceil``)."""
    s = (int(h * ratio), int(w * ratio))
    if not same_shape:
        h, w = (math.ceil(x * ratio / gs) * gs for x in (h, w))
    return s, (h, w)


def scale_img(img, ratio=1.0, same_shape=False, gs=32, flip=None):
    """reference torch_utils.py:262-288 on the device: bilinear resize (align_corners=False) of ``img`` [B, C, H, W] to
    ``int(h * ratio) x int(w * ratio)``, padded bottom / right with 0.447 to a multiple of ``gs`` (or back to (h, w) with
    ``same_shape``).  ``flip`` (None, 2 or 3) folds the caller's ``img.flip(flip)`` into the same pass.  A uint8 image is
    divided by 255 first (validate.py:236-238); the result is always float32.  ``ratio == 1.0`` without flip returns ``img``."""
    from .metrics import _handle
    if not (torch.is_tensor(img) and img.is_cuda and img.dim() == 4):
        raise N.SkyEyeNativeError("scale_img: img must be a [B, C, H, W] tensor on the HIP device (no CPU path)")
    flip = int(flip or 0)
    if ratio == 1.0 and not flip:
        return img
    if img.dtype != torch.uint8:
        img = img.float()
    src = img.contiguous()
    B, C, H, W = src.shape
    s, p = ((H, W), (H, W)) if ratio == 1.0 else scale_img_geometry(H, W, ratio, same_shape, gs)
    out = torch.empty((B, C, p[0], p[1]), dtype=torch.float32, device=img.device)
    h = _handle(img.device.index or 0)
    stream = torch.cuda.current_stream(img.device).cuda_stream
    N.check(h.L.sky_scale_img(h.h, src.data_ptr(), N.SKY_IO_U8 if src.dtype == torch.uint8 else N.SKY_IO_F32, B, C, H, W, out.data_ptr(),
                              s[0], s[1], p[0], p[1], flip, 0.447, ctypes.c_void_p(stream)), h.h)
    return out


def capture_graph(fn, warmup=2):
    """Capture ``fn()`` -- a static-shape chain of engine calls such as forward + ``nms_raw``, no host synchronisation inside --
    into ONE hipGraph: ``graph, outputs = capture_graph(step); graph.replay()`` re-runs all its launches (77 for skyeye_s plus
    NMS) with a single host call and refreshes ``outputs`` in place.  The warm-up runs on the capture stream first, so that every
    one-time allocation (plans, NMS workspace, LDS attributes) happens before the capture starts.  Inputs captured by ``fn``
    must keep their storage; to feed new frames copy them into the captured input tensor."""
    if not torch.cuda.is_available():
        raise N.SkyEyeNativeError("capture_graph needs the HIP device (no CPU path)")
    from . import metrics as _metrics
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    keep = []
    _metrics._KEEP.append(keep)
    try:
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            outputs = fn()
    finally:
        _metrics._KEEP.pop()
    # the graph replays into the NMS workspace of the utility handle(s) it was captured with: they live as long as the graph; the
    # capture stream dies with this call, so its table entry goes (a new stream may get the same pointer)
    graph._sky_keep = keep
    _metrics.forget_stream(torch.cuda.current_device(), side.cuda_stream)
    return graph, outputs
