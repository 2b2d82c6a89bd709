"""Comparators shared by the CPU (oracle vs golden) and GPU (engine vs golden / oracle) parity tests."""
import numpy as np

DEFAULT_ANCHORS = [[[10, 13], [16, 30], [33, 23]], [[30, 61], [62, 45], [59, 119]], [[116, 90], [156, 198], [373, 326]]]


def det_close(det, ref, scales, tol=1e-4):
    """Parity criterion for decoded detections [.., N, nc+5] (DESIGN.md "Parity"): tol relative to each
    column's natural scale --
       xy    |d| <= tol * max(|ref|, stride)            xy = (2s - 0.5 + grid) * stride      (detector.py:137)
       wh    |d| <= tol * max(|ref|, anchor * stride)   wh = (2s)^2 * (anchor * stride)      (detector.py:138)
       probs |d| <= tol
    plus IoU(box, box_ref) >= 1 - 10*tol for boxes of at least 8 px, and identical class argmax.
    Any two fp32 implementations of this 75-107 layer graph differ by ~2e-5 relative in the logits (different
    summation order per convolution), which is what these scales absorb."""
    d = np.abs(det.astype(np.float64) - ref.astype(np.float64))
    lim = np.full(ref.shape, tol, dtype=np.float64)
    lim[..., 0:2] = tol * np.maximum(np.abs(ref[..., 0:2]), scales[..., 0:1])
    lim[..., 2:4] = tol * np.maximum(np.abs(ref[..., 2:4]), scales[..., 1:3])
    bad = d > lim
    assert not bad.any(), f"{bad.sum()} elements out of tolerance, worst ratio {(d / lim).max():.2f}"

    def corners(b):
        b = b.astype(np.float64)
        return b[..., 0] - b[..., 2] / 2, b[..., 1] - b[..., 3] / 2, b[..., 0] + b[..., 2] / 2, b[..., 1] + b[..., 3] / 2

    ax1, ay1, ax2, ay2 = corners(det)
    bx1, by1, bx2, by2 = corners(ref)
    iw = np.clip(np.minimum(ax2, bx2) - np.maximum(ax1, bx1), 0, None)
    ih = np.clip(np.minimum(ay2, by2) - np.maximum(ay1, by1), 0, None)
    inter = iw * ih
    union = (ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter
    sized = (ref[..., 2] >= 8.0) & (ref[..., 3] >= 8.0)
    iou = np.where(sized, inter / np.maximum(union, 1e-30), 1.0)
    assert iou.min() >= 1 - 10 * tol, f"min IoU {iou.min()}"
    if det.shape[-1] > 6:
        assert np.array_equal(det[..., 5:].argmax(-1), ref[..., 5:].argmax(-1)), "class indices differ"


def level_scales(hw, anchors=None, strides=(8, 16, 32), grids=None):
    """per detection row: [stride, anchor_w * stride, anchor_h * stride] (anchor*stride quirk, detector.py:119-121)"""
    anchors = anchors or DEFAULT_ANCHORS
    rows = []
    for i, (lvl, st) in enumerate(zip(anchors, strides)):
        gh, gw = grids[i] if grids is not None else (hw[0] // st, hw[1] // st)
        for a in lvl:
            rows.append(np.tile(np.asarray([[st, a[0] * st, a[1] * st]], dtype=np.float64), (gh * gw, 1)))
    return np.concatenate(rows, 0)


def close(a, b, rtol=2e-5):
    """dense tensors: max |a - b| <= rtol * max(1, max|b|)"""
    assert a.shape == b.shape, f"{a.shape} vs {b.shape}"
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())
    assert err <= rtol * scale, f"max err {err:.3e} > {rtol * scale:.3e}"
