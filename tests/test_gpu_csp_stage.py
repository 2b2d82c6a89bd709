"""The fused first CSP stage (k_csp_stage.hip: cv1|cv2 -> bottleneck 1x1 -> bottleneck 3x3 + shortcut -> cv3 in one launch, every
intermediate map stays in LDS / registers) against the four-launch form (SKY_NO_CSP_STAGE=1): bit-identical -- same K order, same
rounding points -- on whole backbones and detectors, ragged sizes (partial tiles, borders inside tiles), B = 32 at 1280 x 1280,
and deterministic.  The four-launch form is what test_gpu_blocks.py / test_gpu_detector.py pin against the reference fixtures."""
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
        os.environ["SKY_NO_CSP_STAGE"] = "1"
    try:
        outs = m._run([x])
        h = m._engine([m._prepare_input(x)])
        info = [h.op_info(i) for i in range(h.stats()["launches"])]
    finally:
        os.environ.pop("SKY_NO_CSP_STAGE", None)
    return outs, info


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 96, 160), (2, 200, 136), (3, 256, 320), (1, 640, 640), (1, 72, 104)], ids=lambda s: "b%d_%dx%d" % s)
def test_csp_stage_equals_four_launch_form(shape):
    B, H, W = shape
    x = torch.from_numpy(seeded_scene(B, H, W, 33)).cuda()
    of, info_f = _backbone(x, True)
    ou, info_u = _backbone(x, False)
    assert sum("csp-stage-fused" in t for t in info_f) == 1, info_f[:8]
    assert sum("fused-into-previous" in t for t in info_f) >= 3, info_f[:8]
    assert not any("csp-stage-fused" in t for t in info_u)
    for a, b in zip(of, ou):
        assert bool(torch.isfinite(a).all())
        assert torch.equal(a, b), f"{int((a != b).sum())} of {a.numel()} values differ, max {float((a - b).abs().max())}"


def test_csp_block_alone_matches_the_unfused_block_and_the_oracle():
    """CSPBlock(64, 64, 1) as a module: the stage kernel is what runs, against the four launches and the CPU oracle."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import skyeye_oracle as O
    from helpers import seeded_state_for
    from seeded import seeded_input
    m = load_seeded(M.CSPBlock(64, 64, 1), 41).set_precision("bf16")
    P = seeded_state_for(m, 41)
    x = seeded_input("csp.stage.x", (2, 64, 40, 56), 9, -1.0, 1.0)
    xg = torch.from_numpy(x).cuda()
    y = m(xg).cpu().numpy()
    os.environ["SKY_NO_CSP_STAGE"] = "1"
    try:
        y4 = m(xg).cpu().numpy()
    finally:
        os.environ.pop("SKY_NO_CSP_STAGE", None)
    assert np.array_equal(y, y4)
    ref = O.csp(P, "", x)
    assert float(np.abs(y - ref).max()) <= 3e-2 * max(1.0, float(np.abs(ref).max()))
    # without the shortcut (the neck's flavour of the block): fused == four launches
    m2 = load_seeded(M.CSPBlock(64, 64, 1, shortcut=False), 42).set_precision("bf16")
    y = m2(xg).cpu().numpy()
    os.environ["SKY_NO_CSP_STAGE"] = "1"
    try:
        y4 = m2(xg).cpu().numpy()
    finally:
        os.environ.pop("SKY_NO_CSP_STAGE", None)
    assert np.array_equal(y, y4) and np.isfinite(y).all()


def test_detector_b32_1280_fused_equals_unfused_and_is_deterministic():
    P = detector_params("skyeye_s")
    x = torch.from_numpy(seeded_scene(32, 1280, 1280, 92)).cuda()
    dets = {}
    for fused in (True, False):
        m = build_detector(variant_cfg("skyeye_s"))
        m.load_state_dict({k: torch.from_numpy(np.asarray(a)) for k, a in P.items()}, strict=True)
        m.eval().set_precision("bf16")
        if not fused:
            os.environ["SKY_NO_CSP_STAGE"] = "1"
        try:
            d1, _ = m(x, return_raw=False)
            d2, _ = m(x, return_raw=False)
        finally:
            os.environ.pop("SKY_NO_CSP_STAGE", None)
        assert torch.equal(d1, d2)
        dets[fused] = d1
    assert torch.equal(dets[True], dets[False]), f"{int((dets[True] != dets[False]).sum())} values differ"
