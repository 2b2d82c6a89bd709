"""BottleneckBlock(128, 128) as ONE kernel -- round 4's two 4-wave workgroups per CU on 8 x 16 tiles (k_bneck_w.hip, the default) and round 3's
one 8-wave workgroup per CU on 16 x 16 tiles (k_bneck.hip: SKY_BNECK128=solo; cv1 on the halo tile, u kept in LDS, weights through a four-stage
ring with counted waits): both bit-identical to the two-launch form (SKY_NO_BNECK128=1: 1x1 on the streaming kernel + 3x3 with residual on the
halo-tile kernel) -- same MFMA instructions in the same K order, same bf16 roundings -- on CSP blocks with 2, 3 and 4 bottlenecks,
ragged maps (image borders inside tiles), the 80 x 80 size of the detector at B = 32 (several tiles per workgroup, the ring running
across tile boundaries) and the single-tile / last-tile cases; deterministic; close to the fp32 engine."""
import os

import pytest
import torch

import skyeye.core.models as M
from helpers import load_seeded
from seeded import seeded_input

pytestmark = pytest.mark.gpu

CASES = [(3, 2, 48, 48), (2, 1, 16, 16), (2, 2, 40, 56), (4, 1, 33, 47), (3, 32, 80, 80), (3, 2, 24, 100), (2, 3, 160, 160)]


def _run(n, x, fused, shortcut=True, solo=False):
    m = load_seeded(M.CSPBlock(256, 256, num_blocks=n, shortcut=shortcut), 23).set_precision("bf16")
    if not fused:
        os.environ["SKY_NO_BNECK128"] = "1"
    if solo:
        os.environ["SKY_BNECK128"] = "solo"
    os.environ["SKY_CONV_HALO"] = "force"         # ragged maps: both forms take the halo-tile kernels whatever the tile fill
    try:
        y = m(x)
        h = m._engine([x])
        info = [h.op_info(i) for i in range(h.stats()["launches"])]
    finally:
        os.environ.pop("SKY_NO_BNECK128", None)
        os.environ.pop("SKY_BNECK128", None)
        os.environ.pop("SKY_CONV_HALO", None)
    return y, info


@pytest.mark.parametrize("case", CASES, ids=["n%d_b%d_%dx%d" % c for c in CASES])
def test_bneck128_equals_two_launch_form(case):
    n, B, H, W = case
    x = torch.from_numpy(seeded_input("bk128.x.%d.%d" % (H, W), (B, 256, H, W), 3, -2.0, 2.0)).cuda()
    yf, info_f = _run(n, x, True)
    yu, info_u = _run(n, x, False)
    assert sum("bneck128" in t for t in info_f) == n, info_f              # the fused kernel really ran, once per bottleneck
    assert not any("bneck128" in t for t in info_u) and len(info_u) == len(info_f) + n, info_u
    assert bool(torch.isfinite(yf).all())
    assert torch.equal(yf, yu), f"{int((yf != yu).sum())} of {yf.numel()} values differ, max {float((yf - yu).abs().max())}"
    yf2, _ = _run(n, x, True)
    assert torch.equal(yf, yf2)
    assert sum("bneck128x2" in t for t in info_f) == n, info_f             # the default is the two-workgroups-per-CU kernel ...
    ys, info_s = _run(n, x, True, solo=True)                                # ... and round 3's kernel gives the same bits
    assert sum("bneck128" in t for t in info_s) == n and not any("bneck128x2" in t for t in info_s), info_s
    assert torch.equal(ys, yu)


def test_bneck128_without_shortcut():
    x = torch.from_numpy(seeded_input("bk128.ns", (2, 256, 40, 40), 7, -2.0, 2.0)).cuda()
    yf, info_f = _run(2, x, True, shortcut=False)
    yu, _ = _run(2, x, False, shortcut=False)
    assert sum("bneck128" in t for t in info_f) == 2
    assert torch.equal(yf, yu)


def test_bneck128_against_fp32_engine():
    x = torch.from_numpy(seeded_input("bk128.ref", (2, 256, 64, 64), 5, -2.0, 2.0)).cuda()
    yf, _ = _run(3, x, True)
    ref = load_seeded(M.CSPBlock(256, 256, num_blocks=3), 23).set_precision("fp32")(x)
    err = float((yf - ref).abs().max() / ref.abs().max())
    assert err < 0.03, err


# ---- fp8 engine (round 4): k_bneck_w8.hip -- 128 channels are one 128-byte chunk, 16x16x128 block-scaled instructions -------------------------
# The two-launch form it is compared with keeps the fused plan's buffers (SKY_BNECK128=pair: y1 -> A -> B -> y1 through the scratch tensors, the hidden
# tensor materialised where the fused plan has its scale carrier), so both plans calibrate to the same scales; the in-place unfused plan
# (SKY_NO_BNECK128=1) shares ONE scale between the stages of a CSP block and is only close.
def _run8(n, x, mode, shortcut=True, width=256):
    m = load_seeded(M.CSPBlock(width, width, num_blocks=n, shortcut=shortcut), 23).set_precision("fp8")
    if mode == "pair":
        os.environ["SKY_BNECK128"] = "pair"
    if mode == "big":                             # the 16 x 16-tile form of the kernel (measured slower, kept as a switch)
        os.environ["SKY_BNECK128"] = "solo"
    if mode == "inplace":
        os.environ["SKY_NO_BNECK128"] = "1"
    os.environ["SKY_CONV_HALO"] = "force"
    try:
        y = m(x)
        h = m._engine([x])
        info = [h.op_info(i) for i in range(h.stats()["launches"])]
        scales = h.scales()
    finally:
        os.environ.pop("SKY_NO_BNECK128", None)
        os.environ.pop("SKY_BNECK128", None)
        os.environ.pop("SKY_CONV_HALO", None)
    return y, info, scales


CASES8 = [(3, 2, 48, 48), (2, 1, 16, 16), (2, 2, 40, 56), (4, 1, 33, 47), (3, 16, 96, 96), (3, 2, 24, 100)]


@pytest.mark.parametrize("case", CASES8, ids=["n%d_b%d_%dx%d" % c for c in CASES8])
def test_fp8_bneck128_equals_two_launch_form(case):
    n, B, H, W = case
    x = torch.from_numpy(seeded_input("bk128f8.x.%d.%d" % (H, W), (B, 256, H, W), 3, -2.0, 2.0)).cuda()
    yf, info_f, sc_f = _run8(n, x, "fused")
    yp, info_p, sc_p = _run8(n, x, "pair")
    assert sum("bneck128x2-fp8" in t for t in info_f) == n, info_f
    assert not any("bneck128" in t for t in info_p) and len(info_p) == len(info_f) + n, info_p
    assert len(sc_f) == len(sc_p) and all(a == b for a, b in zip(sc_f, sc_p)), "the two plans calibrated to different scales"
    assert bool(torch.isfinite(yf).all()) and float(yf.abs().max()) > 0
    assert torch.equal(yf, yp), f"{int((yf != yp).sum())} of {yf.numel()} values differ, max {float((yf - yp).abs().max())}"
    yf2, _, _ = _run8(n, x, "fused")
    assert torch.equal(yf, yf2)
    yb, info_b, _ = _run8(n, x, "big")
    assert sum("bneck128x2-fp8" in t for t in info_b) == n and torch.equal(yb, yp)


def test_fp8_bneck128_without_shortcut_and_against_the_in_place_plan():
    x = torch.from_numpy(seeded_input("bk128f8.ns", (2, 256, 40, 40), 7, -2.0, 2.0)).cuda()
    yf, info_f, _ = _run8(2, x, "fused", shortcut=False)
    yp, _, _ = _run8(2, x, "pair", shortcut=False)
    assert sum("bneck128x2-fp8" in t for t in info_f) == 2 and torch.equal(yf, yp)
    # the in-place plan (one scale for all stages of the block) is another quantisation of the same network: close, not equal
    yf3, _, _ = _run8(3, x, "fused")
    yi, info_i, _ = _run8(3, x, "inplace")
    assert not any("bneck128" in t for t in info_i)
    ref = load_seeded(M.CSPBlock(256, 256, num_blocks=3), 23).set_precision("fp32")(x)
    ef = float((yf3.float() - ref).norm() / ref.norm()), float((yi.float() - ref).norm() / ref.norm())
    print("fp8 CSP(256, n=3) relative L2 against fp32: fused plan %.4f, in-place plan %.4f" % ef)
    assert ef[0] < 0.12 and ef[1] < 0.12, ef


# BottleneckBlock(64, 64) of the fp8 engine (k_bneck_w64f8.hip): 64-byte pixels, the nine taps paired into four 16x16x128 instructions + one 16x16x32
# pair exactly as the narrow halo kernel pairs its K-steps
CASES64 = [(3, 2, 48, 48), (2, 1, 16, 16), (2, 2, 40, 56), (4, 1, 33, 47), (3, 8, 192, 192), (3, 2, 24, 100)]


@pytest.mark.parametrize("case", CASES64, ids=["n%d_b%d_%dx%d" % c for c in CASES64])
def test_fp8_bneck64_equals_two_launch_form(case):
    n, B, H, W = case
    x = torch.from_numpy(seeded_input("bk64f8.x.%d.%d" % (H, W), (B, 128, H, W), 3, -2.0, 2.0)).cuda()
    yf, info_f, sc_f = _run8(n, x, "fused", width=128)
    yp, info_p, sc_p = _run8(n, x, "pair", width=128)
    assert sum("bneck64x4-fp8" in t for t in info_f) == n, info_f
    assert not any("bneck64" in t for t in info_p) and len(info_p) == len(info_f) + n, info_p
    assert len(sc_f) == len(sc_p) and all(a == b for a, b in zip(sc_f, sc_p)), "the two plans calibrated to different scales"
    assert bool(torch.isfinite(yf).all()) and float(yf.abs().max()) > 0
    assert torch.equal(yf, yp), f"{int((yf != yp).sum())} of {yf.numel()} values differ, max {float((yf - yp).abs().max())}"
    yf2, _, _ = _run8(n, x, "fused", width=128)
    assert torch.equal(yf, yf2)


def test_fp8_bneck64_without_shortcut():
    x = torch.from_numpy(seeded_input("bk64f8.ns", (2, 128, 40, 40), 7, -2.0, 2.0)).cuda()
    yf, info_f, _ = _run8(2, x, "fused", shortcut=False, width=128)
    yp, _, _ = _run8(2, x, "pair", shortcut=False, width=128)
    assert sum("bneck64x4-fp8" in t for t in info_f) == 2 and torch.equal(yf, yp)
