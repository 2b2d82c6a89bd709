"""The neck's cat([up2(lateral(p)), q]) is not materialised for its upsampled half: the CSP's fused cv1 | cv2 GEMM (streaming kernel)
reads those channels of K from the small lateral map at pixel (y >> 1, x >> 1) (ConvArgs::in2).  Same values in the same K order:
FeatureNeck and the whole detector are bit-identical to the materialised form (SKY_NO_IN2=1), in bf16 and in the exact fp32 engine, on
even and on odd maps (odd ones keep the upsample launch), B = 8 @320 and the detector's 1280 size."""
import os

import numpy as np
import pytest
import torch

from skyeye.core.models.detector import FeatureNeck
from helpers import build_detector, detector_params, load_seeded, variant_cfg
from seeded import seeded_input, seeded_scene

pytestmark = pytest.mark.gpu


def _neck(prec, feats, small):
    if not small:
        os.environ["SKY_NO_IN2"] = "1"
    try:
        m = load_seeded(FeatureNeck([128, 256, 512], width_multiple=1.0), 31).set_precision(prec)
        outs = m(feats)
        h = m._engine(list(feats))
        info = [h.op_info(i) for i in range(h.stats()["launches"])]
    finally:
        os.environ.pop("SKY_NO_IN2", None)
    return outs, info


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
@pytest.mark.parametrize("hw", [(40, 40), (24, 56), (20, 36)])
def test_neck_reads_the_small_lateral_map(prec, hw):
    B, (H5, W5) = 2, hw
    feats = [torch.from_numpy(seeded_input("in2.p%d.%d.%d" % (i, H5, W5), (B, c, H5 * s, W5 * s), 3 + i, -2.0, 2.0)).cuda()
             for i, (c, s) in enumerate([(128, 4), (256, 2), (512, 1)])]
    a, info_a = _neck(prec, feats, True)
    b, info_b = _neck(prec, feats, False)
    assert sum(" in2" in t for t in info_a) == 2, info_a
    assert not any(" in2" in t for t in info_b)
    assert sum(" up2" in t for t in info_b) == 2 and not any(" up2" in t for t in info_a)
    for x, y in zip(a, b):
        assert torch.equal(x, y), f"{int((x != y).sum())} of {x.numel()} values differ"


def test_detector_with_and_without_the_materialised_concat():
    P = detector_params("skyeye_s")

    def run(small, x):
        if not small:
            os.environ["SKY_NO_IN2"] = "1"
        try:
            m = build_detector(variant_cfg("skyeye_s"))
            m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()}, strict=True)
            d, raw = m.eval().set_precision("bf16")(x)
        finally:
            os.environ.pop("SKY_NO_IN2", None)
        return d, raw

    for B, S in ((8, 320), (1, 1280), (2, 96)):
        x = torch.from_numpy(seeded_scene(B, S, S if S != 96 else 160, 41)).cuda()
        d0, r0 = run(True, x)
        d1, r1 = run(False, x)
        assert torch.equal(d0, d1) and all(torch.equal(p, q) for p, q in zip(r0, r1))


# fp8 (round 4): one multiplier per output channel covers both parts of K because the small map carries the concat buffer's scale (tied); the
# calibration twin writes the concat anyway, so both plans measure the same ranges and the bytes are equal
@pytest.mark.parametrize("hw", [(20, 20), (12, 28), (10, 18)])
def test_fp8_neck_reads_the_small_lateral_map(hw):
    B, (H5, W5) = 2, hw
    feats = [torch.from_numpy(seeded_input("in2f8.p%d.%d.%d" % (i, H5, W5), (B, c, H5 * s, W5 * s), 3 + i, -2.0, 2.0)).cuda()
             for i, (c, s) in enumerate([(256, 4), (512, 2), (1024, 1)])]

    def neck(small):
        if not small:
            os.environ["SKY_NO_IN2"] = "1"
        try:
            m = load_seeded(FeatureNeck([256, 512, 1024], width_multiple=1.0), 31).set_precision("fp8")
            outs = m(feats)
            h = m._engine(list(feats))
            return outs, [h.op_info(i) for i in range(h.stats()["launches"])], h.scales()
        finally:
            os.environ.pop("SKY_NO_IN2", None)

    a, info_a, sc_a = neck(True)
    b, info_b, sc_b = neck(False)
    assert sum(" in2" in t for t in info_a) == 2, info_a
    assert not any(" in2" in t for t in info_b) and sum(" up2" in t for t in info_b) == 2
    for x, y in zip(a, b):
        assert bool(torch.isfinite(x.float()).all()) and torch.equal(x, y), f"{int((x != y).sum())} of {x.numel()} values differ"


def test_fp8_detector_with_and_without_the_materialised_concat():
    """skyeye_l (256 / 512 / 1024-channel levels: lateral maps of whole 256-byte slabs; skyeye_s's 64 / 128 bytes stay materialised)."""
    P = detector_params("skyeye_l")

    def run(small, x):
        if not small:
            os.environ["SKY_NO_IN2"] = "1"
        try:
            m = build_detector(variant_cfg("skyeye_l"))
            m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()}, strict=True)
            m = m.eval().set_precision("fp8")
            d, raw = m(x)
            h = m._engine([m._prepare_input(x)])
            return d, raw, sum(" in2" in h.op_info(i) for i in range(h.stats()["launches"]))
        finally:
            os.environ.pop("SKY_NO_IN2", None)

    for B, H, W in ((2, 320, 320), (1, 640, 384)):
        x = torch.from_numpy(seeded_scene(B, H, W, 41)).cuda()
        d0, r0, n0 = run(True, x)
        d1, r1, n1 = run(False, x)
        assert (n0, n1) == (2, 0)
        assert torch.equal(d0, d1) and all(torch.equal(p, q) for p, q in zip(r0, r1))
