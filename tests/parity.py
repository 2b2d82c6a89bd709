"""Comparators shared by the CPU (oracle vs golden) and GPU (engine vs golden / oracle) parity tests."""
import inspect
import json
import os

import numpy as np

# Every comparison records its margin (worst error / limit, minimum IoU): tests/conftest.py prints the table at the end of
# the run and writes gpurun_out/parity_margins.json, so that the slack of each parity case is visible, not just pass/fail.
MARGINS = []
IOU_TOL = 1e-4        # matched boxes of at least 8 px: IoU >= 1 - IOU_TOL


def _label():
    for fr in inspect.stack()[2:8]:
        if os.path.basename(fr.filename).startswith("test_") or fr.function == "smoke":
            loc = fr.frame.f_locals
            case = loc.get("case") if isinstance(loc.get("case"), dict) else None
            name = loc.get("name") or (case or {}).get("name") or ""
            extra = loc.get("prec") or loc.get("precision") or ""
            return f"{os.path.basename(fr.filename)}::{fr.function}[{name}{'/' + str(extra) if extra else ''}]"
    return "?"


def write_margins(path):
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(MARGINS, f, indent=1)
    except OSError:
        pass


DEFAULT_ANCHORS = [[[10, 13], [16, 30], [33, 23]], [[30, 61], [62, 45], [59, 119]], [[116, 90], [156, 198], [373, 326]]]


def det_close(det, ref, scales, tol=1e-4, iou_tol=None):
    """Parity criterion for decoded detections [.., N, nc+5] (DESIGN.md "Parity"): tol relative to each
    column's natural scale --
       xy    |d| <= tol * max(|ref|, stride)            xy = (2s - 0.5 + grid) * stride      (detector.py:137)
       wh    |d| <= tol * max(|ref|, anchor * stride)   wh = (2s)^2 * (anchor * stride)      (detector.py:138)
       probs |d| <= tol
    plus IoU(box, box_ref) >= 1 - iou_tol (default IOU_TOL = 1e-4, scaled with tol) for boxes of at least 8 px, and
    identical class argmax.  Returns and records the margins: worst |d| / limit ratio, minimum IoU.
    Any two fp32 implementations of this 75-107 layer graph differ by ~2e-5 relative in the logits (different
    summation order per convolution), which is what these scales absorb."""
    d = np.abs(det.astype(np.float64) - ref.astype(np.float64))
    lim = np.full(ref.shape, tol, dtype=np.float64)
    lim[..., 0:2] = tol * np.maximum(np.abs(ref[..., 0:2]), scales[..., 0:1])
    lim[..., 2:4] = tol * np.maximum(np.abs(ref[..., 2:4]), scales[..., 1:3])
    bad = d > lim
    worst = float((d / lim).max()) if d.size else 0.0
    rec = dict(where=_label(), kind="det", worst_ratio=round(worst, 4), tol=tol, rows=int(np.prod(det.shape[:-1])))
    MARGINS.append(rec)
    assert not bad.any(), f"{bad.sum()} elements out of tolerance, worst ratio {worst:.2f}"

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
    if iou_tol is None:
        iou_tol = IOU_TOL * tol / 1e-4
    rec["one_minus_min_iou"] = float(1.0 - iou.min()) if iou.size else 0.0
    rec["iou_tol"] = iou_tol
    assert iou.min() >= 1 - iou_tol, f"min IoU {iou.min()} (1 - {1 - iou.min():.3e}), limit 1 - {iou_tol:.1e}"
    if det.shape[-1] > 6:
        same = np.array_equal(det[..., 5:].argmax(-1), ref[..., 5:].argmax(-1))
        rec["class_equal"] = bool(same)
        assert same, "class indices differ"
    return rec


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
    rec = dict(where=_label(), kind="dense", worst_ratio=round(err / (rtol * scale), 4), tol=rtol, max_err=err, scale=scale)
    MARGINS.append(rec)
    assert err <= rtol * scale, f"max err {err:.3e} > {rtol * scale:.3e}"
    return rec


# ---- reduced-precision engines (bf16, fp8): agreement rates, recorded like the margins above ---------------------------------
def e4m3_table():
    """OCP e4m3fn value of every byte (0x7f / 0xff = NaN)."""
    t = np.empty(256, np.float32)
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 15, b & 7
        v = np.nan if (e == 15 and m == 7) else (m * 2.0 ** -9 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 7))
        t[b] = -v if s else v
    return t


def quantize_e4m3(x):
    """float array -> nearest e4m3fn VALUE (round to nearest even, saturating at +-448), as the engine's epilogues do."""
    t = e4m3_table()
    pos = np.sort(t[:127])                                           # 0 .. 448, the 127 non-negative finite values
    a = np.clip(np.abs(np.asarray(x, np.float64)), 0, 448.0)
    hi = np.clip(np.searchsorted(pos, a, side="left"), 1, 126)
    lo = hi - 1
    dlo, dhi = a - pos[lo], pos[hi] - a
    pick_hi = (dhi < dlo) | ((dhi == dlo) & (hi % 2 == 0))          # even index = even mantissa (table order = code order)
    q = np.where(pick_hi, pos[hi], pos[lo])
    return (np.sign(x) * q).astype(np.float32)


def box_agreement(a, b, iou_thr=0.9):
    """Post-NMS rows a, b ([n, >= 6]: x1, y1, x2, y2, conf, cls): fraction of b's boxes that have a box of the same class in a with
    IoU >= iou_thr (each box of a used once, best IoU first), and the mean IoU of the matched pairs."""
    if len(b) == 0:
        return 1.0, 1.0
    if len(a) == 0:
        return 0.0, 0.0
    ax1, ay1, ax2, ay2 = a[:, 0:1], a[:, 1:2], a[:, 2:3], a[:, 3:4]
    bx1, by1, bx2, by2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    iw = np.clip(np.minimum(ax2, bx2) - np.maximum(ax1, bx1), 0, None)
    ih = np.clip(np.minimum(ay2, by2) - np.maximum(ay1, by1), 0, None)
    inter = iw * ih
    iou = inter / np.maximum((ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter, 1e-9)
    iou = np.where(a[:, 5:6] == b[:, 5], iou, 0.0)
    used_a, used_b, ious = set(), set(), []
    for k in np.argsort(-iou, axis=None):
        i, j = divmod(int(k), iou.shape[1])
        if iou[i, j] < iou_thr:
            break
        if i in used_a or j in used_b:
            continue
        used_a.add(i); used_b.add(j); ious.append(iou[i, j])
    return len(used_b) / len(b), float(np.mean(ious)) if ious else 0.0


def record_agreement(name, **rates):
    MARGINS.append(dict(where=_label(), kind="agreement", case=name, worst_ratio=0.0, **{k: round(float(v), 5) for k, v in rates.items()}))


def row_iou(det, ref, mask):
    """IoU of the decoded boxes (cx, cy, w, h) of the SAME rows (same cell and anchor) of two engines, over the rows in `mask`."""
    a, b = det[mask].astype(np.float64), ref[mask].astype(np.float64)
    if len(a) == 0:
        return np.ones(0)
    ax1, ay1, ax2, ay2 = a[:, 0] - a[:, 2] / 2, a[:, 1] - a[:, 3] / 2, a[:, 0] + a[:, 2] / 2, a[:, 1] + a[:, 3] / 2
    bx1, by1, bx2, by2 = b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2
    inter = np.clip(np.minimum(ax2, bx2) - np.maximum(ax1, bx1), 0, None) * np.clip(np.minimum(ay2, by2) - np.maximum(ay1, by1), 0, None)
    return inter / np.maximum((ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter, 1e-9)
