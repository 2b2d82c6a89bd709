"""The few helpers of reference skyeye/utils/general.py that touch the inference path, plus the box utilities the
reference's CLIs import from it but never define (detect.py:24-26, validate.py:23-25: scale_boxes, xywh2xyxy,
xyxy2xywh).  Small host-side arithmetic on at most max_det boxes per image -- not part of the HIP hot path."""
import math

import torch


def make_divisible(x, divisor):
    """reference general.py:234-245"""
    return math.ceil(x / divisor) * divisor


def check_img_size(img_size, stride=32, s=None):
    """reference general.py:248-268 (callers pass ``s=``: validate.py:188, detect.py:111 -- both spellings accepted)."""
    if s is not None:
        stride = s
    stride = int(stride.max() if torch.is_tensor(stride) else stride)
    if isinstance(img_size, int):
        new = max(make_divisible(img_size, stride), stride)
    else:
        new = [max(make_divisible(x, stride), stride) for x in img_size]
    if new != img_size:
        print(f"WARNING: --img-size {img_size} must be multiple of max stride {stride}, updating to {new}")
    return new


def xywh2xyxy(x):
    y = x.clone()
    y[..., 0] = x[..., 0] - x[..., 2] / 2
    y[..., 1] = x[..., 1] - x[..., 3] / 2
    y[..., 2] = x[..., 0] + x[..., 2] / 2
    y[..., 3] = x[..., 1] + x[..., 3] / 2
    return y


def xyxy2xywh(x):
    y = x.clone()
    y[..., 0] = (x[..., 0] + x[..., 2]) / 2
    y[..., 1] = (x[..., 1] + x[..., 3]) / 2
    y[..., 2] = x[..., 2] - x[..., 0]
    y[..., 3] = x[..., 3] - x[..., 1]
    return y


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None):
    """Map xyxy boxes from the letterboxed shape back to the original image (inverse of augmentation.py:442-496:
    gain = min(new/old), symmetric padding), clipped to the image."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    boxes = boxes.clone()
    boxes[..., [0, 2]] -= pad[0]
    boxes[..., [1, 3]] -= pad[1]
    boxes[..., :4] /= gain
    boxes[..., [0, 2]] = boxes[..., [0, 2]].clamp(0, img0_shape[1])
    boxes[..., [1, 3]] = boxes[..., [1, 3]].clamp(0, img0_shape[0])
    return boxes
