"""Deterministic, name-keyed synthetic weights and inputs (numpy only).

Shared by ``make_golden.py`` (which pushes them into the reference's PyTorch
classes through ``load_state_dict``) and by the tests / bench (which push the
same arrays into the HIP engine and the CPU oracle).  Every tensor depends only
on ``(seed, name, shape)`` so the three consumers never have to agree on an
iteration order.

BatchNorm statistics, affine terms and biases are randomised so that BN folding,
bias handling and the attention MLPs are actually exercised (the reference's
``_initialize_weights``, skyeye/core/models/detector.py:326-341, would leave BN at
identity and every bias at zero, hiding folding mistakes).
"""
import zlib

import numpy as np


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode("utf-8"))])


def seeded_tensor(name, shape, seed=0):
    """float32 (int64 for ``num_batches_tracked``) array for a state-dict entry."""
    shape = tuple(int(s) for s in shape)
    r = _rng(seed, name)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if leaf == "running_var":
        return r.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == "relative_position_bias_table":
        return (0.5 * r.standard_normal(shape)).astype(np.float32)
    if leaf in ("weight", "in_proj_weight"):
        if len(shape) == 4:
            # conv: fan-in scaled normal.  The reference's own init N(0, sqrt(2/(k*k*cout))) (detector.py:331-333)
            # makes activations grow ~1e6x through the 100+ layers of skyeye_l with random BN statistics, which
            # turns every comparison into a test of overflow; gain 1.3 keeps all three model sizes O(1)
            # (gain >= 1.4 diverges).  Detection layers get gain 1.2: logits with std ~2.5, so that the ~2e-5 relative difference any two fp32
            # implementations of a 100-layer network show stays below 1e-4 after the sigmoid decode.
            gain = 1.2 if "detection_layers" in name else 1.3
            fan_in = shape[1] * shape[2] * shape[3]
            return (gain / np.sqrt(fan_in) * r.standard_normal(shape)).astype(np.float32)
        if len(shape) == 2:  # linear: fan-in scaled so attention MLPs are not ~0
            return (r.standard_normal(shape) / np.sqrt(shape[1])).astype(np.float32)
        if len(shape) == 1:  # BN / LN gamma
            return r.uniform(0.8, 1.2, shape).astype(np.float32)
    if leaf in ("bias", "in_proj_bias"):
        if ".bn." in name or name.startswith("bn."):
            # BatchNorm beta around +1.5: SiLU then works mostly in its near-linear range, which keeps a 100-layer
            # random network well-conditioned (with beta ~ 0 rounding differences are amplified ~1e3x by depth
            # and even two fp32 CPU implementations disagree at 1e-3)
            return (1.5 + 0.3 * r.standard_normal(shape)).astype(np.float32)
        scale = 1.0 if "detection_layers" in name else 0.1
        return (scale * r.standard_normal(shape)).astype(np.float32)
    raise ValueError(f"seeded_tensor: no rule for {name!r} with shape {shape}")


def seeded_state(spec, seed=0):
    """spec: iterable of (name, shape) -> {name: ndarray}."""
    return {name: seeded_tensor(name, shape, seed) for name, shape in spec}


def seeded_input(name, shape, seed=0, lo=0.0, hi=1.0):
    """float32 uniform [lo, hi) tensor for activations / feature maps."""
    r = _rng(seed, "input:" + name)
    return r.uniform(lo, hi, tuple(shape)).astype(np.float32)


def seeded_frames(batch, height, width, seed=0):
    """uint8 frames [B,3,H,W], i.i.d. uniform 0..255 (SURVEY 8d / BASELINE.md 4)."""
    r = np.random.default_rng(int(seed))
    return r.integers(0, 256, size=(batch, 3, height, width), dtype=np.uint8)


def seeded_scene(batch, height, width, seed=0):
    """uint8 frames [B,3,H,W] with structure at every scale (blocky value noise over 6 octaves + rectangles +
    pixel noise).  Used for the parity fixtures and BatchNorm calibration: on i.i.d. pixel noise the deep feature
    maps of a random network are almost constant, BatchNorm then divides by a vanishing variance and the whole
    graph becomes chaotic, which tests nothing but overflow."""
    r = np.random.default_rng([int(seed), 0x5CE7E])
    out = np.empty((batch, 3, height, width), dtype=np.uint8)
    ys, xs = np.arange(height), np.arange(width)
    for b in range(batch):
        img = np.zeros((3, height, width), dtype=np.float64)
        amp = 1.0
        for s in (3, 5, 9, 17, 33, 65):
            g = r.uniform(-1.0, 1.0, (3, s, s))
            img += amp * g[:, (ys * s // height)[:, None], (xs * s // width)[None, :]]
            amp *= 0.7
        for _ in range(10):
            y0, x0 = int(r.integers(0, height)), int(r.integers(0, width))
            hh, ww = int(r.integers(2, max(3, height // 3))), int(r.integers(2, max(3, width // 3)))
            img[:, y0:y0 + hh, x0:x0 + ww] += r.uniform(-1.5, 1.5, (3, 1, 1))
        img += 0.15 * r.standard_normal(img.shape)
        lo, hi = img.min(), img.max()
        out[b] = np.clip((img - lo) / (hi - lo) * 255.0, 0, 255).astype(np.uint8)
    return out
