"""The large-K 1x1 GEMM (csrc/k_gemm1x1.hip: 256-pixel x 256-channel tiles, both operands by LDS-DMA through a two-stage ring of 128-byte
rows) against the streaming kernel it replaces (SKY_NO_GEMM1X1=1): same MFMA instruction, operand roles and K order, so
ConvolutionBlock(k = 1) (reference blocks.py:10-41), the neck with its in-place concat (ConvArgs::in2, detector.py:210-229) and whole
detectors (also the head-attention variant: 1x1 with a residual, 1536 / 2048 output rows) are bit-identical -- on ragged pixel counts (tiles past M, waves without pixels), one and several N tiles, K from 192 to 1024
channels, SiLU and no activation, and at the benchmark's B = 32 shapes where the GEMM is the default path."""
import os

import numpy as np
import pytest
import torch

from skyeye.core.models import ConvolutionBlock
from skyeye.core.models.detector import FeatureNeck
from helpers import build_detector, detector_params, load_seeded, variant_cfg
from seeded import seeded_input, seeded_scene

pytestmark = pytest.mark.gpu


def _env(gemm, force):
    os.environ.pop("SKY_NO_GEMM1X1", None)
    os.environ.pop("SKY_GEMM1X1", None)
    if not gemm:
        os.environ["SKY_NO_GEMM1X1"] = "1"
    elif force:
        os.environ["SKY_GEMM1X1"] = "force"


def _clear():
    os.environ.pop("SKY_NO_GEMM1X1", None)
    os.environ.pop("SKY_GEMM1X1", None)


def _conv(cin, cout, act, x, gemm, force=True, precision="bf16"):
    _env(gemm, force)
    try:
        m = load_seeded(ConvolutionBlock(cin, cout, 1, 1, activation=act), 17).eval().set_precision(precision)
        y = m(x)
        h = m._engine([x])
        info = [h.op_info(i) for i in range(h.stats()["launches"])]
    finally:
        _clear()
    return y, info


@pytest.mark.parametrize("cin,cout,shape,act", [
    (512, 512, (2, 40, 40), True), (256, 256, (1, 13, 9), True), (1024, 512, (1, 20, 24), True), (192, 256, (3, 7, 5), True),
    (768, 512, (2, 17, 23), False), (512, 256, (2, 80, 80), True), (384, 256, (1, 31, 33), True), (256, 1024, (1, 12, 20), True),
    (192, 512, (1, 128, 1), True)], ids=lambda v: str(v).replace(" ", ""))
def test_gemm1x1_equals_stream_kernel(cin, cout, shape, act):
    B, H, W = shape
    x = torch.from_numpy(seeded_input("g1.%d.%d.%d.%d" % (cin, B, H, W), (B, cin, H, W), 5, -2.0, 2.0)).cuda()
    y_g, info_g = _conv(cin, cout, act, x, True)
    y_s, info_s = _conv(cin, cout, act, x, False)
    assert any("gemm1x1" in t for t in info_g), info_g
    assert not any("gemm1x1" in t for t in info_s), info_s
    assert bool(torch.isfinite(y_g).all())
    assert torch.equal(y_g, y_s), f"{int((y_g != y_s).sum())} of {y_g.numel()} values differ"


# fp8 (round 4): 128 channels per 128-byte row and step, one 16x16x128 block-scaled instruction per row pair -- the streaming kernel's own pairing
# of K-steps (fp8_mma128) and its epilogue (acc * multiplier + bias, SiLU, 1 / out scale, e4m3), so the bytes are equal
@pytest.mark.parametrize("cin,cout,shape,act", [
    (512, 512, (2, 40, 40), True), (384, 256, (1, 13, 9), True), (1024, 512, (1, 20, 24), True), (768, 512, (2, 17, 23), False),
    (512, 256, (2, 80, 80), True), (384, 1024, (1, 12, 20), True), (640, 256, (1, 128, 1), True)], ids=lambda v: str(v).replace(" ", ""))
def test_fp8_gemm1x1_equals_stream_kernel(cin, cout, shape, act):
    B, H, W = shape
    x = torch.from_numpy(seeded_input("g1f8.%d.%d.%d.%d" % (cin, B, H, W), (B, cin, H, W), 5, -2.0, 2.0)).cuda()
    y_g, info_g = _conv(cin, cout, act, x, True, precision="fp8")
    y_s, info_s = _conv(cin, cout, act, x, False, precision="fp8")
    assert any("gemm1x1" in t for t in info_g), info_g
    assert not any("gemm1x1" in t for t in info_s), info_s
    assert bool(torch.isfinite(y_g).all()) and float(y_g.abs().max()) > 0
    assert torch.equal(y_g, y_s), f"{int((y_g != y_s).sum())} of {y_g.numel()} values differ"


def test_fp8_gemm1x1_refuses_what_it_cannot_tile():
    """K below three 128-channel steps or not a multiple of 128 stays on the streaming kernel even when forced."""
    for cin in (256, 192, 320):
        x = torch.from_numpy(seeded_input("g1f8.no.%d" % cin, (1, cin, 16, 16), 5, -2.0, 2.0)).cuda()
        _, info = _conv(cin, 256, True, x, True, precision="fp8")
        assert not any("gemm1x1" in t for t in info), info


def test_gemm1x1_default_choice():
    """Cin >= 512, Cout >= 512 and half a tile per CU or more: the GEMM without the force switch; small maps stay on the streaming kernel."""
    x = torch.from_numpy(seeded_input("g1.big", (32, 512, 40, 40), 6, -2.0, 2.0)).cuda()
    y_g, info = _conv(512, 512, True, x, True, force=False)
    assert any("gemm1x1" in t for t in info), info
    y_s, _ = _conv(512, 512, True, x, False)
    assert torch.equal(y_g, y_s)
    x = torch.from_numpy(seeded_input("g1.small", (1, 512, 20, 20), 6, -2.0, 2.0)).cuda()
    _, info = _conv(512, 512, True, x, True, force=False)
    assert not any("gemm1x1" in t for t in info), info


@pytest.mark.parametrize("hw", [(40, 40), (24, 56), (20, 36)])
def test_neck_second_input_through_the_gemm(hw):
    B, (H5, W5) = 2, hw
    feats = [torch.from_numpy(seeded_input("g1n.p%d.%d.%d" % (i, H5, W5), (B, c, H5 * s, W5 * s), 3 + i, -2.0, 2.0)).cuda()
             for i, (c, s) in enumerate([(128, 4), (256, 2), (512, 1)])]

    def neck(gemm):
        _env(gemm, True)
        try:
            m = load_seeded(FeatureNeck([128, 256, 512], width_multiple=1.0), 31).set_precision("bf16")
            outs = m(feats)
            h = m._engine(list(feats))
            return outs, [h.op_info(i) for i in range(h.stats()["launches"])]
        finally:
            _clear()

    a, info_a = neck(True)
    b, info_b = neck(False)
    assert any(" in2" in t and "gemm1x1" in t for t in info_a), info_a
    assert not any("gemm1x1" in t for t in info_b)
    for x, y in zip(a, b):
        assert torch.equal(x, y), f"{int((x != y).sum())} of {x.numel()} values differ"


@pytest.mark.parametrize("hw", [(40, 40), (20, 36)])
def test_fp8_neck_second_input_through_the_gemm(hw):
    B, (H5, W5) = 2, hw
    feats = [torch.from_numpy(seeded_input("g1n8.p%d.%d.%d" % (i, H5, W5), (B, c, H5 * s, W5 * s), 3 + i, -2.0, 2.0)).cuda()
             for i, (c, s) in enumerate([(256, 4), (512, 2), (1024, 1)])]

    def neck(gemm):
        _env(gemm, True)
        try:
            m = load_seeded(FeatureNeck([256, 512, 1024], width_multiple=1.0), 31).set_precision("fp8")
            outs = m(feats)
            h = m._engine(list(feats))
            return outs, [h.op_info(i) for i in range(h.stats()["launches"])]
        finally:
            _clear()

    a, info_a = neck(True)
    b, info_b = neck(False)
    assert any(" in2" in t and "gemm1x1" in t for t in info_a), info_a
    assert not any("gemm1x1" in t for t in info_b)
    for x, y in zip(a, b):
        assert torch.equal(x, y), f"{int((x != y).sum())} of {x.numel()} values differ"


def test_fp8_detector_with_and_without_the_gemm():
    """skyeye_l (BASELINE.json configs[4] / [5]'s model) in fp8, the GEMM forced onto every 1x1 layer it can tile (its default choice needs the
    benchmark's pixel counts): same detections, bit for bit, also through the neck's in-place concat (the calibration twin materialises it)."""
    P = detector_params("skyeye_l")
    x = torch.from_numpy(seeded_scene(3, 256, 320, 43)).cuda()

    def run(gemm):
        _env(gemm, True)
        try:
            m = build_detector(variant_cfg("skyeye_l"))
            m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()}, strict=True)
            m = m.eval().set_precision("fp8")
            d, raw = m(x)
            h = m._engine([m._prepare_input(x)])
            return d, raw, [h.op_info(i) for i in range(h.stats()["launches"])]
        finally:
            _clear()

    d0, r0, i0 = run(True)
    d1, r1, i1 = run(False)
    assert sum("gemm1x1" in t for t in i0) >= 4 and any(" in2" in t and "gemm1x1" in t for t in i0), i0
    assert not any("gemm1x1" in t for t in i1) and any(" in2" in t for t in i1), i1
    assert torch.equal(d0, d1) and all(torch.equal(p, q) for p, q in zip(r0, r1))


@pytest.mark.parametrize("variant,shape", [("skyeye_s", (2, 320, 320)), ("skyeye_s", (1, 1280, 1280)), ("skyeye_l", (2, 128, 96)), ("skyeye_s", (3, 96, 160)),
                                           ("skyeye_s_ha", (3, 256, 384)), ("skyeye_s_ha", (1, 1280, 1280))],
                         ids=lambda v: v if isinstance(v, str) else "b%d_%dx%d" % v)
def test_detector_with_and_without_the_gemm(variant, shape):
    P = detector_params(variant)
    B, H, W = shape
    x = torch.from_numpy(seeded_scene(B, H, W, 43)).cuda()

    def run(gemm):
        _env(gemm, True)
        try:
            m = build_detector(variant_cfg(variant))
            m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()}, strict=True)
            m = m.eval().set_precision("bf16")
            d, raw = m(x)
            h = m._engine([m._prepare_input(x)])
            return d, raw, [h.op_info(i) for i in range(h.stats()["launches"])]
        finally:
            _clear()

    d0, r0, i0 = run(True)
    d1, r1, i1 = run(False)
    assert sum("gemm1x1" in t for t in i0) >= 8, i0
    if variant == "skyeye_s_ha":        # TransformerLayer: out_proj / FFN-down with the residual in the epilogue, QKV / FFN-up with up to 2048 rows
        assert sum("gemm1x1" in t and "+res" in t for t in i0) >= 2, i0
        assert any("->2048" in t and "gemm1x1" in t for t in i0), i0
    assert not any("gemm1x1" in t for t in i1)
    assert torch.equal(d0, d1) and all(torch.equal(p, q) for p, q in zip(r0, r1))
