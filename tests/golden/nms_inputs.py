"""Synthetic ``[B, N, nc+5]`` prediction tensors for the NMS wrapper cases.

The reference wrapper (skyeye/utils/metrics.py:361-457) treats columns 0-3 as
corner coordinates without converting them, so half of the
rows are drawn with ``col2 > col0`` and ``col3 > col1`` inside a handful of
clusters -- that is what makes suppression actually happen in literal mode.
"""
import numpy as np


def make_predictions(nc, batch, n, seed, ties=False, distinct_scores=False):
    r = np.random.default_rng(int(seed))
    pred = np.empty((batch, n, nc + 5), dtype=np.float32)
    for b in range(batch):
        k = 12
        centres = r.uniform(0.0, 200.0, (k, 2))
        sizes = r.uniform(220.0, 420.0, (k, 2))
        which = r.integers(0, k, n)
        jitter = r.normal(0.0, 6.0, (n, 4))
        box = np.concatenate([centres[which], sizes[which]], axis=1) + jitter
        # second half: generic cx,cy,w,h rows (mostly degenerate as corners)
        half = n // 2
        box[half:, 0:2] = r.uniform(0.0, 640.0, (n - half, 2))
        box[half:, 2:4] = r.uniform(4.0, 160.0, (n - half, 2))
        obj = r.beta(0.6, 1.6, n)
        if distinct_scores:
            obj = (r.permutation(n) + 0.5) / n
        if ties:
            obj = np.round(obj * 8.0) / 8.0
        cls = r.beta(0.7, 1.4, (n, nc))
        if ties:
            cls = np.round(cls * 4.0) / 4.0
        # cluster members share similar class confidences so the literal
        # "offset by cls_conf*4096" (metrics.py:438) still leaves overlaps
        base = r.uniform(0.3, 0.95, (k, nc))
        cls[:half] = np.clip(base[which[:half]] + r.choice([0.0, 0.0, 0.0, 0.01], (half, nc)), 0, 1)
        pred[b, :, 0:4] = box
        pred[b, :, 4] = obj
        pred[b, :, 5:] = cls
    return pred
