"""Deep-pipelined 3x3 convolution (k_conv3x3_deep.hip: one workgroup of 8 waves per CU, two halo images, four-stage weight ring with
counted waits) for inputs of 256 channels and more: bit-identical to the halo-tile kernel (SKY_NO_DEEP3X3=1) -- same MFMA
instructions in the same K order, same roundings -- as a plain ConvolutionBlock and as cv2 of bottlenecks (residual), on ragged maps,
several work items per workgroup (B = 16 @80 x 80), 512 channels (two passes of the step body, four N tiles), single-item launches;
deterministic; close to a PyTorch fp32 convolution."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import skyeye.core.models as M
from helpers import load_seeded
from seeded import seeded_input

pytestmark = pytest.mark.gpu


def _run(make, x, deep):
    m = load_seeded(make(), 29).set_precision("bf16")
    if not deep:
        os.environ["SKY_NO_DEEP3X3"] = "1"
    os.environ["SKY_CONV_HALO"] = "force"         # ragged maps: both forms take their tile kernels whatever the tile fill
    try:
        y = m(x)
        h = m._engine([x])
        info = [h.op_info(i) for i in range(h.stats()["launches"])]
    finally:
        os.environ.pop("SKY_NO_DEEP3X3", None)
        os.environ.pop("SKY_CONV_HALO", None)
    return y, info, m


CONV_CASES = [(256, 256, 2, 48, 48), (256, 128, 1, 16, 16), (256, 256, 2, 40, 56), (512, 512, 1, 33, 47), (256, 256, 16, 80, 80), (512, 256, 2, 32, 32),
              (1024, 128, 1, 24, 24)]


@pytest.mark.parametrize("case", CONV_CASES, ids=["%d-%d_b%d_%dx%d" % c for c in CONV_CASES])
def test_deep3x3_conv_equals_halo_kernel(case):
    cin, cout, B, H, W = case
    x = torch.from_numpy(seeded_input("deep.x.%d.%d.%d" % (cin, H, W), (B, cin, H, W), 3, -2.0, 2.0)).cuda()
    yd, info_d, m = _run(lambda: M.ConvolutionBlock(cin, cout, 3, 1), x, True)
    yh, info_h, _ = _run(lambda: M.ConvolutionBlock(cin, cout, 3, 1), x, False)
    assert any("deep3x3" in t for t in info_d), info_d
    assert not any("deep3x3" in t for t in info_h), info_h
    assert bool(torch.isfinite(yd).all())
    assert torch.equal(yd, yh), f"{int((yd != yh).sum())} of {yd.numel()} values differ, max {float((yd - yh).abs().max())}"
    yd2, _, _ = _run(lambda: M.ConvolutionBlock(cin, cout, 3, 1), x, True)
    assert torch.equal(yd, yd2)
    # against PyTorch on the same bf16-rounded operands (BatchNorm folded by hand)
    sd = {k: v.float() for k, v in m.state_dict().items()}
    s = sd["bn.weight"] / torch.sqrt(sd["bn.running_var"] + 1e-5)
    w = (sd["conv.weight"] * s[:, None, None, None]).bfloat16().float().cuda()
    b = (sd["bn.bias"] - sd["bn.running_mean"] * s).cuda()
    ref = F.silu(F.conv2d(x.bfloat16().float(), w, b, padding=1))
    err = float((yd - ref).abs().max() / ref.abs().max())
    assert err < 0.02, err


@pytest.mark.parametrize("case", [(2, 2, 48, 48), (3, 8, 80, 80), (2, 1, 24, 40)], ids=["n2_b2_48", "n3_b8_80", "n2_b1_24x40"])
def test_deep3x3_as_bottleneck_cv2_with_residual(case):
    n, B, H, W = case
    x = torch.from_numpy(seeded_input("deep.csp.%d.%d" % (H, W), (B, 512, H, W), 5, -2.0, 2.0)).cuda()
    yd, info_d, _ = _run(lambda: M.CSPBlock(512, 512, num_blocks=n), x, True)
    yh, info_h, _ = _run(lambda: M.CSPBlock(512, 512, num_blocks=n), x, False)
    assert sum("deep3x3" in t and "+res" in t for t in info_d) == n, info_d
    assert torch.equal(yd, yh), f"{int((yd != yh).sum())} of {yd.numel()} values differ, max {float((yd - yh).abs().max())}"
