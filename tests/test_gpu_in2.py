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
