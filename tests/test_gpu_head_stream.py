"""The streaming detection-level kernel (csrc/k_head.hip: 1x1 convolution + bias + decode of DetectionHead.forward /
process_detections, reference detector.py:61-145, pixels straight from global memory, weights resident in LDS) against the
implicit-GEMM tile kernel's detection epilogue (SKY_NO_HEAD_STREAM=1): bit-identical detections and raw levels -- same K order, same
epilogue arithmetic -- on whole detectors (all three levels: 128 / 256 / 512 input channels for skyeye_s, 256 / 512 / 1024 for
skyeye_l), ragged sizes (fragments that straddle images, pixels past M), with and without the raw levels."""
import os

import numpy as np
import pytest
import torch

from helpers import build_detector, detector_params, variant_cfg
from seeded import seeded_scene

pytestmark = pytest.mark.gpu


def _detector(variant):
    m = build_detector(variant_cfg(variant))
    m.load_state_dict({k: torch.from_numpy(np.asarray(a)) for k, a in detector_params(variant).items()}, strict=True)
    return m.eval().set_precision("bf16")


def _run(m, x, stream, return_raw=True, force=True, cv3_head=False):
    if not cv3_head:
        os.environ["SKY_NO_CV3_HEAD"] = "1"           # (the cv3 + level kernel has its own tests below)
    if not stream:
        os.environ["SKY_NO_HEAD_STREAM"] = "1"
    elif force:
        os.environ["SKY_HEAD_STREAM"] = "force"       # small levels too (by default they stay on the tile kernel)
    try:
        det, raw = m(x, return_raw=return_raw)
        torch.cuda.synchronize()
        h = m._engine([m._prepare_input(x)])
        info = [h.op_info(i) for i in range(h.stats()["launches"])]
        return det, raw, info
    finally:
        os.environ.pop("SKY_NO_HEAD_STREAM", None)
        os.environ.pop("SKY_HEAD_STREAM", None)
        os.environ.pop("SKY_NO_CV3_HEAD", None)


@pytest.mark.parametrize("variant,shape", [("skyeye_s", (2, 64, 64)), ("skyeye_s", (3, 96, 160)), ("skyeye_s", (1, 96, 224)),
                                           ("skyeye_s", (5, 32, 32)), ("skyeye_l", (2, 128, 96)), ("skyeye_s", (2, 640, 640))],
                         ids=lambda v: v if isinstance(v, str) else "b%d_%dx%d" % v)
def test_head_stream_equals_tile_kernel(variant, shape):
    B, H, W = shape
    m = _detector(variant)
    x = torch.from_numpy(seeded_scene(B, H, W, 57)).cuda()
    det_s, raw_s, info_s = _run(m, x, True)
    det_t, raw_t, info_t = _run(m, x, False)
    assert sum("head-stream" in t for t in info_s) == 3, [t for t in info_s if "head" in t]
    assert not any("head-stream" in t for t in info_t)
    assert bool(torch.isfinite(det_s).all())
    assert torch.equal(det_s, det_t), f"{int((det_s != det_t).sum())} of {det_s.numel()} detections differ"
    for a, b in zip(raw_s, raw_t):
        assert torch.equal(a, b)
    # without the raw levels (the bench path): same detections
    det_n, raw_n, _ = _run(m, x, True, return_raw=False)
    assert torch.equal(det_n, det_s)


def test_head_stream_b32_1280_is_deterministic_and_equals_tile_kernel():
    m = _detector("skyeye_s")
    x = torch.from_numpy(seeded_scene(32, 1280, 1280, 58)).cuda()
    d1, _, info = _run(m, x, True, return_raw=False, force=False)
    d2, _, _ = _run(m, x, True, return_raw=False, force=False)
    d3, _, _ = _run(m, x, False, return_raw=False)
    assert sum("head-stream" in t for t in info) == 2, [t for t in info if "head" in t]     # P3 and P4; P5 (51 200 pixels) on the tile kernel
    assert torch.equal(d1, d2) and torch.equal(d1, d3)


# ---- fpn_conv3.cv3 + detection level 0 in one kernel (cv3_head_kernel) ------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 64, 64), (3, 96, 160), (1, 96, 224), (5, 32, 32), (2, 640, 640), (1, 224, 288)],
                         ids=lambda v: "b%d_%dx%d" % v)
def test_cv3_head_equals_two_launches(shape):
    """skyeye_s: P3 has 128 channels, so the CSP cv3 that writes it and level 0 pair up.  The fused kernel's P3 map feeds the PAN path
    (levels 1 and 2 depend on it), so equal detections on every level also pin the map."""
    B, H, W = shape
    m = _detector("skyeye_s")
    x = torch.from_numpy(seeded_scene(B, H, W, 61)).cuda()
    det_f, raw_f, info_f = _run(m, x, True, cv3_head=True)
    det_s, raw_s, info_s = _run(m, x, True)
    assert sum("cv3+head" in t for t in info_f) == 1, [t for t in info_f if "head" in t]
    assert sum("fused-into-previous" in t and " head" in t for t in info_f) == 1
    assert not any("cv3+head" in t for t in info_s)
    assert bool(torch.isfinite(det_f).all())
    assert torch.equal(det_f, det_s), f"{int((det_f != det_s).sum())} of {det_f.numel()} detections differ"
    for a, b in zip(raw_f, raw_s):
        assert torch.equal(a, b)
    det_n, _, _ = _run(m, x, True, return_raw=False, cv3_head=True)
    assert torch.equal(det_n, det_f)


def test_cv3_head_b32_1280_default_path():
    m = _detector("skyeye_s")
    x = torch.from_numpy(seeded_scene(32, 1280, 1280, 58)).cuda()
    d1, _, info = _run(m, x, True, return_raw=False, force=False, cv3_head=True)
    d2, _, _ = _run(m, x, True, return_raw=False, force=False, cv3_head=True)
    d3, _, _ = _run(m, x, True, return_raw=False, force=False)
    assert sum("cv3+head" in t for t in info) == 1, [t for t in info if "head" in t]
    assert torch.equal(d1, d2) and torch.equal(d1, d3)


def test_cv3_head_not_taken_where_it_does_not_apply():
    """skyeye_l's P3 has 256 channels: the pairing is not made; small levels without the force switch stay on two launches."""
    m = _detector("skyeye_l")
    x = torch.from_numpy(seeded_scene(2, 128, 96, 57)).cuda()
    _, _, info = _run(m, x, True, cv3_head=True)
    assert not any("cv3+head" in t for t in info)
    m = _detector("skyeye_s")
    x = torch.from_numpy(seeded_scene(2, 64, 64, 57)).cuda()
    _, _, info = _run(m, x, True, force=False, cv3_head=True)
    assert not any("cv3+head" in t for t in info)


def test_cv3_head_pairs_the_output_projection_of_the_head_attention_variant():
    """skyeye_s_ha: level 0 reads the window attention's output projection (1x1 128 -> 128, no activation): the same pairing, bit-identical."""
    m = _detector("skyeye_s_ha")
    x = torch.from_numpy(seeded_scene(3, 256, 384, 80)).cuda()
    det_f, raw_f, info_f = _run(m, x, True, cv3_head=True)
    det_s, raw_s, info_s = _run(m, x, True)
    assert sum("cv3+head" in t for t in info_f) == 1, [t for t in info_f if "head" in t or "128->128" in t]
    assert not any("cv3+head" in t for t in info_s)
    assert torch.equal(det_f, det_s)
    for a, b in zip(raw_f, raw_s):
        assert torch.equal(a, b)
