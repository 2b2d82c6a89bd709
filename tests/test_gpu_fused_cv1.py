"""The fused BottleneckBlock(64, 64) launch -- round 4's three 4-wave workgroups per CU on 8 x 16 tiles (k_bneck_w64.hip, the default) and the halo-tile
kernel's form (k_conv_halo.hip CV1, SKY_NO_BNECK64W=1: cv1 computed on the 18 x 18 halo tile of the 3x3, residual from the LDS tile): both
bit-identical to the two-launch form (SKY_NO_FUSE_CV1=1) -- same MFMA instructions in the same order, same bf16 roundings --
on CSP blocks with 2, 3 and 4 bottlenecks, ragged maps (image borders inside tiles, non-square tile shapes) and B = 16 at the
160 x 160 size of the detector (two workgroups per CU, several tiles per workgroup), and deterministic."""
import os

import numpy as np
import pytest
import torch

import skyeye.core.models as M
from helpers import load_seeded
from seeded import seeded_input

pytestmark = pytest.mark.gpu

CASES = [(3, 2, 48, 48), (2, 2, 40, 56), (4, 1, 33, 47), (3, 16, 160, 160), (3, 2, 24, 100)]


def _run(n, x, fused, halo=False, force=False):
    m = load_seeded(M.CSPBlock(128, 128, num_blocks=n), 19).set_precision("bf16")
    if not fused:
        os.environ["SKY_NO_FUSE_CV1"] = "1"
    if halo:
        os.environ["SKY_NO_BNECK64W"] = "1"
    if force:
        os.environ["SKY_CONV_HALO"] = "force"       # ragged maps: whatever the tile fill
    try:
        y = m(x)
        h = m._engine([x])
        info = [h.op_info(i) for i in range(h.stats()["launches"])]
    finally:
        os.environ.pop("SKY_NO_FUSE_CV1", None)
        os.environ.pop("SKY_NO_BNECK64W", None)
        os.environ.pop("SKY_CONV_HALO", None)
    return y, info


@pytest.mark.parametrize("case", CASES, ids=["n%d_b%d_%dx%d" % c for c in CASES])
def test_fused_bottlenecks_equal_two_launch_form(case):
    n, B, H, W = case
    x = torch.from_numpy(seeded_input("cv1.x.%d.%d" % (H, W), (B, 128, H, W), 3, -2.0, 2.0)).cuda()
    yf, info_f = _run(n, x, True, halo=True)
    yu, info_u = _run(n, x, False)
    assert sum("halo-cv1+3x3" in t for t in info_f) == n, info_f          # the fused kernel really ran, once per bottleneck
    assert not any("halo-cv1+3x3" in t or "bneck64" in t for t in info_u) and len(info_u) == len(info_f) + n
    assert bool(torch.isfinite(yf).all())
    assert torch.equal(yf, yu), f"{int((yf != yu).sum())} of {yf.numel()} values differ, max {float((yf - yu).abs().max())}"
    yf2, _ = _run(n, x, True, halo=True)
    assert torch.equal(yf, yf2)
    yw, info_w = _run(n, x, True, force=True)                              # the default: k_bneck_w64.hip
    assert sum("bneck64x3" in t for t in info_w) == n, info_w
    assert torch.equal(yw, yu), f"{int((yw != yu).sum())} of {yw.numel()} values differ, max {float((yw - yu).abs().max())}"
    yw2, _ = _run(n, x, True, force=True)
    assert torch.equal(yw, yw2)


def test_bneck64w_without_shortcut_and_single_tiles():
    for (B, H, W), sc in (((2, 40, 40), False), ((1, 8, 16), True), ((3, 7, 13), True), ((1, 160, 16), True)):
        x = torch.from_numpy(seeded_input("cv1w.%d.%d" % (H, W), (B, 128, H, W), 3, -2.0, 2.0)).cuda()
        m = load_seeded(M.CSPBlock(128, 128, num_blocks=2, shortcut=sc), 19).set_precision("bf16")
        outs = []
        for env in ({"SKY_CONV_HALO": "force"}, {"SKY_NO_FUSE_CV1": "1"}):
            os.environ.update(env)
            try:
                outs.append(m(x))
                h = m._engine([x])
                info = [h.op_info(i) for i in range(h.stats()["launches"])]
            finally:
                for k in env:
                    os.environ.pop(k, None)
            if "SKY_CONV_HALO" in env:
                assert sum("bneck64x3" in t for t in info) == 2, info
        assert torch.equal(outs[0], outs[1]), (B, H, W, sc)


def test_fused_bottleneck_against_fp32_engine():
    x = torch.from_numpy(seeded_input("cv1.ref", (2, 128, 64, 64), 5, -2.0, 2.0)).cuda()
    yf, _ = _run(3, x, True)
    ref = load_seeded(M.CSPBlock(128, 128, num_blocks=3), 19).set_precision("fp32")(x)
    err = float((yf - ref).abs().max() / ref.abs().max())
    assert err < 0.03, err
