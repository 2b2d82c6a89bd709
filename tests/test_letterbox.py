"""letterbox (reference core/data/augmentation.py:442-496).  CPU: the geometry arithmetic on hand
cases and the oracle's resize on identities; GPU: sky_letterbox bit for bit against the oracle.  The resize itself is
PARITY UNPINNED against OpenCV (cv2 is not installed here and the reference ships no image fixtures)."""
import os

import numpy as np
import pytest
import torch

from skyeye.core.data.augmentation import letterbox, letterbox_geometry


def _oracle():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import skyeye_oracle as O
    return O


def test_geometry_follows_the_reference_arithmetic():
    # 720 x 1280 frame into 640: r = 0.5 -> 640 x 360, auto pads the height to the next multiple of 32 (384): 12 + 12
    ratio, unpad, (dw, dh), (t, b, l, r) = letterbox_geometry((720, 1280), 640, auto=True, stride=32)
    assert ratio == (0.5, 0.5) and unpad == (640, 360) and (dw, dh) == (0.0, 12.0) and (t, b, l, r) == (12, 12, 0, 0)
    # odd total padding: 2.5 per side -> round(2.4) = 2 on top / left, round(2.6) = 3 at the bottom / right (augmentation.py:492-493)
    ratio, unpad, (dw, dh), (t, b, l, r) = letterbox_geometry((100, 59), (64, 64), auto=False)
    assert unpad == (38, 64) and (dw, dh) == (13.0, 0.0) and (l, r) == (13, 13)
    ratio, unpad, (dw, dh), (t, b, l, r) = letterbox_geometry((101, 60), (64, 69), auto=False)
    assert unpad[1] == 64 and (l, r) == (int(round(dw - 0.1)), int(round(dw + 0.1))) and l + r + unpad[0] == 69
    # scaleup=False never enlarges; scale_fill stretches
    assert letterbox_geometry((100, 100), 640, auto=False, scaleup=False)[1] == (100, 100)
    ratio, unpad, pads, _ = letterbox_geometry((100, 200), (64, 64), auto=False, scale_fill=True)
    assert unpad == (64, 64) and ratio == (0.32, 0.64) and pads == (0.0, 0.0)


def test_oracle_resize_identities():
    O = _oracle()
    img = np.random.default_rng(1).integers(0, 256, (37, 53, 3), dtype=np.uint8)
    same = O.letterbox_pixels(img, 37, 53, 2, 3, 4, 5)
    assert same.shape == (42, 62, 3) and np.array_equal(same[2:39, 4:57], img) and (same[:2] == 114).all() and (same[:, 57:] == 114).all()
    flat = np.full((40, 60, 3), 77, np.uint8)
    assert (O.letterbox_pixels(flat, 25, 31, 0, 0, 0, 0) == 77).all()          # bilinear of a constant is the constant
    up = O.letterbox_pixels(img, 74, 106, 0, 0, 0, 0)                          # exact 2x: every output lies between its two taps
    assert up.shape == (74, 106, 3) and up.min() >= img.min() and up.max() <= img.max()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,new_shape,kw", [
    ((720, 1280), 640, dict()),
    ((375, 500), (640, 640), dict(auto=False)),
    ((1080, 1920), (1280, 1280), dict(auto=True)),
    ((97, 61), (128, 128), dict(auto=False, scaleup=True)),
    ((64, 64), (64, 64), dict(auto=False)),
    ((50, 80), (96, 96), dict(auto=False, scale_fill=True)),
], ids=["720p_640_auto", "375x500_640", "1080p_1280_auto", "small_up", "identity", "stretch"])
def test_letterbox_kernel_matches_oracle(shape, new_shape, kw):
    O = _oracle()
    img = np.random.default_rng(shape[0]).integers(0, 256, shape + (3,), dtype=np.uint8)
    out, ratio, pads = letterbox(torch.from_numpy(img).cuda(), new_shape, **kw)
    ratio2, unpad, pads2, (t, b, l, r) = letterbox_geometry(shape, new_shape, **kw)
    assert ratio == ratio2 and pads == pads2
    ref = O.letterbox_pixels(img, unpad[1], unpad[0], t, b, l, r)
    got = out.cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), f"max |diff| {np.abs(got.astype(int) - ref.astype(int)).max()}"
    chw, _, _ = letterbox(torch.from_numpy(img).cuda(), new_shape, chw=True, reverse_channels=True, **kw)
    assert np.array_equal(chw.cpu().numpy(), ref.transpose(2, 0, 1)[::-1])     # detect.py:133
    if isinstance(new_shape, tuple) and not kw.get("auto", True):
        assert got.shape[:2] == new_shape


@pytest.mark.gpu
def test_letterbox_rejects_cpu_tensors():
    with pytest.raises(Exception):
        letterbox(torch.zeros(8, 8, 3, dtype=torch.uint8), 32)
