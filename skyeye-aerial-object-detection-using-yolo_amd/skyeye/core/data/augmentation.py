"""letterbox -- drop-in for reference skyeye/core/data/augmentation.py:442-496 (the resize + pad the inference callers
apply before the model, detect.py:131, and whose inverse is utils.general.scale_boxes), computed on the MI355X.

The geometry (ratio, unpadded size, padding split, the +-0.1 rounding of the borders) is the reference's arithmetic; the
pixels come from ``sky_letterbox`` (OpenCV's 8-bit INTER_LINEAR fixed-point arithmetic + constant border).  cv2 is not
available where this was built and the reference holds no image fixtures: the resize is pinned against oracle/ only."""
import ctypes

import numpy as np
import torch

from ... import _native as N
from ...utils.metrics import _handle


def letterbox_geometry(shape, new_shape=(640, 640), auto=True, scale_fill=False, scaleup=True, stride=32):
    """-> (ratio (w, h), new_unpad (w, h), (dw, dh) halves, (top, bottom, left, right)) exactly as augmentation.py:461-493."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = r, r
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    elif scale_fill:
        dw, dh = 0, 0
        new_unpad = (new_shape[1], new_shape[0])
        ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return ratio, new_unpad, (dw, dh), (top, bottom, left, right)


def letterbox(img, new_shape=(640, 640), color=(114, 114, 114), auto=True, scale_fill=False, scaleup=True, stride=32,
              chw=False, reverse_channels=False):
    """img: uint8 [H, W, 3] tensor on the HIP device -> (letterboxed uint8 image, ratio, (dw, dh)) like the reference.

    ``chw=True`` returns [3, H', W'] (``reverse_channels`` also flips BGR <-> RGB, detect.py:133) ready for the engine."""
    if not (torch.is_tensor(img) and img.is_cuda and img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3):
        raise N.SkyEyeNativeError("letterbox: img must be a uint8 [H, W, 3] tensor on the HIP device (no CPU path)")
    if len(set(int(c) for c in color)) != 1:
        raise NotImplementedError("letterbox: one border value for all channels (the reference's callers use 114)")
    shape = tuple(img.shape[:2])
    ratio, new_unpad, (dw, dh), (top, bottom, left, right) = letterbox_geometry(shape, new_shape, auto, scale_fill, scaleup, stride)
    H1, W1 = new_unpad[1] + top + bottom, new_unpad[0] + left + right
    src = img.contiguous()
    out = torch.empty((3, H1, W1) if chw else (H1, W1, 3), dtype=torch.uint8, device=img.device)
    h = _handle(img.device.index or 0)
    stream = torch.cuda.current_stream(img.device).cuda_stream
    N.check(h.L.sky_letterbox(h.h, src.data_ptr(), shape[0], shape[1], out.data_ptr(), H1, W1, new_unpad[1], new_unpad[0], top, left,
                              int(color[0]), int(chw), int(reverse_channels), ctypes.c_void_p(stream)), h.h)
    return out, ratio, (dw, dh)
