"""The fused stem kernel (k_stem_down.hip: uint8 frames -> FocusBlock convolution -> 3x3 stride-2 convolution, the intermediate
32-channel map never leaves LDS) against the two-launch form (SKY_NO_STEM_DOWN=1): bit-identical -- same K order, same rounding
points -- on whole backbones and detectors, ragged sizes (partial tiles, borders inside tiles), B = 32 at 1280 x 1280, and
deterministic."""
import os

import numpy as np
import pytest
import torch

import skyeye.core.models as M
from helpers import build_detector, detector_params, load_seeded, variant_cfg
from seeded import seeded_scene

pytestmark = pytest.mark.gpu


def _backbone(x, fused):
    m = load_seeded(M.Backbone(base_channels=64, depth_multiple=0.33, width_multiple=0.5), 23).set_precision("bf16")
    if not fused:
        os.environ["SKY_NO_STEM_DOWN"] = "1"
    try:
        outs = m._run([x])
        h = m._engine([m._prepare_input(x)])
        info = [h.op_info(i) for i in range(h.stats()["launches"])]
    finally:
        os.environ.pop("SKY_NO_STEM_DOWN", None)
    return outs, info


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 96, 160), (2, 200, 136), (3, 256, 320), (1, 640, 640)], ids=lambda s: "b%d_%dx%d" % s)
def test_stem_down_equals_two_launch_form(shape):
    B, H, W = shape
    x = torch.from_numpy(seeded_scene(B, H, W, 31)).cuda()
    of, info_f = _backbone(x, True)
    ou, info_u = _backbone(x, False)
    assert sum("stem+stride2-fused" in t for t in info_f) == 1, info_f[:4]
    assert not any("stem+stride2-fused" in t for t in info_u)
    for a, b in zip(of, ou):
        assert bool(torch.isfinite(a).all())
        assert torch.equal(a, b), f"{int((a != b).sum())} of {a.numel()} values differ, max {float((a - b).abs().max())}"


def test_float_frames_keep_the_two_launch_form():
    """The fused kernel reads uint8 frames; float32 frames (the reference's own input contract) take the unfused path and agree."""
    frames = seeded_scene(2, 128, 128, 5)
    a, info_a = _backbone(torch.from_numpy(frames).cuda(), True)
    m = load_seeded(M.Backbone(base_channels=64, depth_multiple=0.33, width_multiple=0.5), 23).set_precision("bf16")
    b = m._run([torch.from_numpy(frames.astype(np.float32) / np.float32(255.0)).cuda()])
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_detector_b32_1280_fused_equals_unfused_and_is_deterministic():
    P = detector_params("skyeye_s")
    x = torch.from_numpy(seeded_scene(32, 1280, 1280, 91)).cuda()
    dets = {}
    for fused in (True, False):
        m = build_detector(variant_cfg("skyeye_s"))
        m.load_state_dict({k: torch.from_numpy(np.asarray(a)) for k, a in P.items()}, strict=True)
        m.eval().set_precision("bf16")
        if not fused:
            os.environ["SKY_NO_STEM_DOWN"] = "1"
        try:
            d1, _ = m(x, return_raw=False)
            d2, _ = m(x, return_raw=False)
        finally:
            os.environ.pop("SKY_NO_STEM_DOWN", None)
        assert torch.equal(d1, d2)
        dets[fused] = d1
    assert torch.equal(dets[True], dets[False]), f"{int((dets[True] != dets[False]).sum())} values differ"
