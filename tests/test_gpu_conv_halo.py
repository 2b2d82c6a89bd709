"""GPU parity of the halo-tile 3x3 convolution kernel (csrc/k_conv_halo.hip) against the CPU oracle on seeded inputs,
with the kernel FORCED on (SKY_CONV_HALO=force) for ragged and small shapes the engine would normally leave to the
streaming kernel, and against the streaming kernel itself.  ConvolutionBlock: reference blocks.py:10-41;
BottleneckBlock (residual in the epilogue): blocks.py:69-90."""
import os

import numpy as np
import pytest
import torch

from helpers import load_seeded, seeded_state_for
from parity import close
from seeded import seeded_input

import skyeye.core.models as M
from skyeye import _native as N

pytestmark = pytest.mark.gpu

# (cin, cout, B, H, W): 128-byte channel chunks = 64 bf16 / 32 fp32 channels
CONV_CASES = [
    (64, 64, 2, 16, 16),      # one tile, one chunk, N_blk 64
    (64, 128, 1, 20, 24),     # ragged tiles, N_blk 128
    (128, 128, 2, 33, 17),    # two chunks, partially filled tiles on both edges
    (128, 64, 3, 16, 48),     # several tiles per image
    (256, 256, 1, 16, 16),    # four chunks, two N tiles
    (192, 192, 1, 9, 40),     # 3 chunks, N_blk 64 x 3
    (64, 64, 5, 32, 32),      # more tiles than one round of workgroups would take on a small grid
    (256, 256, 1, 40, 40),    # 6 x 40 tiles (fragments wrap tile rows), four chunks, two N tiles
    (64, 64, 2, 20, 20),      # 12 x 20 tiles
    (128, 128, 1, 13, 50),    # 5 x 50 tiles, ragged at the bottom
    # narrow inputs (32 or 64 bytes of channels per pixel): resident weights, double-buffered halo
    (16, 32, 2, 20, 24),      # the stem's shape class: bf16 32 B / fp32 64 B per pixel
    (32, 32, 2, 33, 17),      # bf16 64 B per pixel (fp32: the wide kernel)
    (32, 64, 1, 16, 48),
    (16, 64, 3, 16, 16),
    (8, 32, 1, 16, 16),       # fp32 32 B per pixel (bf16: not covered, stays on the streaming kernel)
    (16, 32, 7, 48, 48),      # several tiles per workgroup: both halo buffers in use
    (16, 32, 2, 24, 40),      # 6 x 40 tiles
]


def _oracle():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import skyeye_oracle as O
    return O


def _run(module, x, mode):
    os.environ["SKY_CONV_HALO"] = mode
    try:
        y = module(x)
        torch.cuda.synchronize()
        return y.cpu().numpy()
    finally:
        os.environ.pop("SKY_CONV_HALO", None)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=["%d-%d_b%d_%dx%d" % c for c in CONV_CASES])
def test_halo_conv_matches_oracle(case, prec):
    cin, cout, B, H, W = case
    O = _oracle()
    m = load_seeded(M.ConvolutionBlock(cin, cout, 3, 1), 77).set_precision(prec)
    P = seeded_state_for(m, 77)
    x = seeded_input("halo.x.%d.%d" % (cin, H), (B, cin, H, W), 5, -2.0, 2.0)
    ref = O.conv_block(P, "", x, 3)
    xg = torch.from_numpy(x).cuda()
    y_halo = _run(m, xg, "force")
    y_stream = _run(m, xg, "0")
    if prec == "fp32":
        close(y_halo, ref, rtol=2e-5)          # exact mode: fp32 MFMA, only the summation order differs
        close(y_halo, y_stream, rtol=2e-5)
    else:
        scale = max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(y_halo - ref).max()) <= 4e-2 * scale
        # both kernels round the same bf16 inputs/weights and accumulate in fp32: they may differ by one output ulp
        assert float(np.abs(y_halo - y_stream).max()) <= 2.0 ** -7 * scale


# stride 2: the four parity phases of the input staged one after the other (cin, cout, B, H, W of the INPUT)
S2_CASES = [
    (64, 128, 2, 32, 32),     # one chunk, 16 x 16 output tile
    (128, 256, 1, 33, 41),    # odd input sizes (17 x 21 outputs), two chunks, two N tiles
    (128, 128, 2, 40, 40),    # 20 x 20 outputs
    (256, 64, 1, 16, 16),     # four chunks, N_blk 64
    (64, 64, 3, 64, 64),      # several tiles per workgroup
    # narrow stride-2 kernel (64 bytes of channels per pixel: 32 bf16 / 16 fp32 channels)
    (32, 64, 2, 64, 64),
    (32, 32, 1, 33, 41),
    (16, 64, 2, 32, 48),
    (32, 64, 5, 96, 96),
]


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", S2_CASES, ids=["%d-%d_b%d_%dx%d" % c for c in S2_CASES])
def test_halo_stride2_conv_matches_oracle(case, prec):
    cin, cout, B, H, W = case
    O = _oracle()
    m = load_seeded(M.ConvolutionBlock(cin, cout, 3, 2), 81).set_precision(prec)
    P = seeded_state_for(m, 81)
    x = seeded_input("halo.s2.%d.%d" % (cin, H), (B, cin, H, W), 7, -2.0, 2.0)
    ref = O.conv_block(P, "", x, 3, 2)
    xg = torch.from_numpy(x).cuda()
    y_halo = _run(m, xg, "force")
    y_stream = _run(m, xg, "0")
    assert y_halo.shape == ref.shape
    if prec == "fp32":
        close(y_halo, ref, rtol=2e-5)
        close(y_halo, y_stream, rtol=2e-5)
    else:
        scale = max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(y_halo - ref).max()) <= 4e-2 * scale
        assert float(np.abs(y_halo - y_stream).max()) <= 2.0 ** -7 * scale


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_halo_bottleneck_residual(prec):
    O = _oracle()
    m = load_seeded(M.BottleneckBlock(128, 128, True, 0.5), 78).set_precision(prec)   # cv2: 3x3 64 -> 128 + residual
    P = seeded_state_for(m, 78)
    x = seeded_input("halo.bott", (2, 128, 24, 40), 6, -2.0, 2.0)
    ref = O.bottleneck(P, "", x, True)
    y = _run(m, torch.from_numpy(x).cuda(), "force")
    if prec == "fp32":
        close(y, ref, rtol=2e-5)
    else:
        assert float(np.abs(y - ref).max()) <= 4e-2 * max(1.0, float(np.abs(ref).max()))


def test_halo_kernel_is_what_ran():
    m = load_seeded(M.ConvolutionBlock(64, 128, 3, 1), 79).set_precision("bf16")
    x = torch.randn(2, 64, 32, 32, device="cuda")
    os.environ["SKY_CONV_HALO"] = "force"
    try:
        m(x)
        h = m._engine([x])
        outs = [torch.empty(sh, dtype=torch.float32, device="cuda") for sh in h.output_shapes()]
        prof = h.profile_forward([N.buffer_from_tensor(x)], [N.buffer_from_tensor(t) for t in outs], torch.cuda.current_stream().cuda_stream, iters=1)
    finally:
        os.environ.pop("SKY_CONV_HALO", None)
    tags = [t for _, _, t in prof]
    assert any(t % 10000 == 4128 for t in tags), f"halo kernel did not run: tags {tags}"


def test_stride2_halo_kernel_is_what_ran():
    m = load_seeded(M.ConvolutionBlock(64, 128, 3, 2), 82).set_precision("bf16")
    x = torch.randn(2, 64, 64, 64, device="cuda")
    m(x)
    h = m._engine([x])
    outs = [torch.empty(sh, dtype=torch.float32, device="cuda") for sh in h.output_shapes()]
    prof = h.profile_forward([N.buffer_from_tensor(x)], [N.buffer_from_tensor(t) for t in outs], torch.cuda.current_stream().cuda_stream, iters=1)
    tags = [t for _, _, t in prof]
    assert any(t % 10000 == 6128 for t in tags), f"stride-2 halo kernel did not run: tags {tags}"


def test_narrow_halo_kernel_is_what_ran():
    m = load_seeded(M.ConvolutionBlock(16, 32, 3, 1), 80).set_precision("bf16")
    x = torch.randn(2, 16, 32, 32, device="cuda")
    m(x)
    h = m._engine([x])
    outs = [torch.empty(sh, dtype=torch.float32, device="cuda") for sh in h.output_shapes()]
    prof = h.profile_forward([N.buffer_from_tensor(x)], [N.buffer_from_tensor(t) for t in outs], torch.cuda.current_stream().cuda_stream, iters=1)
    tags = [t for _, _, t in prof]
    assert any(t % 10000 == 5032 for t in tags), f"narrow halo kernel did not run: tags {tags}"
