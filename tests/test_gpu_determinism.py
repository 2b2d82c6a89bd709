"""Large-grid behaviour of the convolution kernels: run-to-run determinism and agreement with a plain PyTorch fp32 convolution
on the same bf16-rounded operands, at batch sizes where every CU holds two workgroups and each workgroup walks several
tiles.  The small parity cases cannot see scheduling-dependent faults: the 128-channel halo tiles passed all of them while a
store-data hazard (register `soffset` on the last vector store of a tile) zeroed a few output vectors per
launch at B = 32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from skyeye.core.models import ConvolutionBlock

pytestmark = pytest.mark.gpu

# cin, cout, k, stride, H = W, batch  -> kernel family taken by the default dispatch
SHAPES = [
    (64, 64, 3, 1, 160, 32),      # halo tile kernel, 64-channel tiles
    (128, 128, 3, 1, 80, 32),     # halo tile kernel, 128-channel tiles
    (256, 256, 3, 1, 40, 32),     # halo tile kernel, non-square tiles
    (32, 32, 3, 1, 320, 16),      # narrow halo kernel
    (64, 128, 3, 2, 320, 16),     # stride 2
    (128, 128, 3, 2, 160, 32),    # stride 2
    (256, 512, 3, 2, 80, 32),     # stride 2, four N tiles
    (128, 128, 1, 1, 160, 32),    # streaming kernel, resident weights
    (512, 512, 1, 1, 40, 32),     # streaming kernel, weight ring
]


@pytest.mark.parametrize("cin,cout,k,stride,hw,batch", SHAPES, ids=[f"{c}to{o}_k{k}s{s}_{h}_b{b}" for c, o, k, s, h, b in SHAPES])
def test_conv_is_deterministic_and_matches_torch(cin, cout, k, stride, hw, batch):
    torch.manual_seed(cin * 1000 + cout + k + stride)
    m = ConvolutionBlock(cin, cout, k, stride).eval().set_precision("bf16")
    x = torch.randn(batch, cin, hw, hw, device="cuda")
    outs = [m(x).clone() for _ in range(3)]
    for o in outs[1:]:
        assert torch.equal(outs[0], o), f"{int((outs[0] != o).sum())} elements differ between two runs of the same launch"
    # reference: BatchNorm (fresh: gamma 1, beta 0, mean 0, var 1) folded, operands rounded to bf16 like the engine's, fp32 math
    w = (m.conv.weight.detach().float().cuda() / np.float32(np.sqrt(1.0 + 1e-5))).bfloat16().float()
    ref = F.silu(F.conv2d(x.bfloat16().float(), w, stride=stride, padding=k // 2))
    got = outs[0]
    assert got.shape == ref.shape
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert err <= 0.02 * scale, f"max |engine - torch| = {err:.4f} at scale {scale:.2f}"      # bf16 output rounding: 2^-9 relative
    lost = int(((got == 0) & (ref.abs() > 0.05 * scale)).sum())
    assert lost == 0, f"{lost} outputs are exactly zero where the reference is not"


def _repeat_equal(fn, n=3):
    outs = [fn().clone() for _ in range(n)]
    for o in outs[1:]:
        assert torch.equal(outs[0], o), f"{int((outs[0] != o).sum())} elements differ between two runs"
    return outs[0]


def test_opt_in_conv_forms_are_deterministic_at_large_grids():
    """The forms the default dispatch does not take -- narrow stride-2 halo kernel (SKY_CONV_HALO=force), the fused 1x1 epilogue
    (SKY_FUSE=1), 64-channel tiles for 128-channel layers (SKY_HALO_NF8=off) -- under the same two-workgroups-per-CU load."""
    import os
    import skyeye.core.models as M
    torch.manual_seed(5)
    x = torch.randn(16, 32, 320, 320, device="cuda")
    os.environ["SKY_CONV_HALO"] = "force"
    try:
        m = ConvolutionBlock(32, 64, 3, 2).eval().set_precision("bf16")
        got = _repeat_equal(lambda: m(x))
    finally:
        os.environ.pop("SKY_CONV_HALO", None)
    w = (m.conv.weight.detach().float().cuda() / np.float32(np.sqrt(1.0 + 1e-5))).bfloat16().float()
    ref = F.silu(F.conv2d(x.bfloat16().float(), w, stride=2, padding=1))
    assert float((got - ref).abs().max()) <= 0.02 * float(ref.abs().max())
    os.environ["SKY_FUSE"] = "1"
    try:
        csp = M.CSPBlock(128, 128, 3, True, 0.5).eval().set_precision("bf16")
        xc = torch.randn(32, 128, 80, 80, device="cuda")
        fused = _repeat_equal(lambda: csp(xc))
    finally:
        os.environ.pop("SKY_FUSE", None)
    assert bool(torch.isfinite(fused).all())
    os.environ["SKY_HALO_NF8"] = "off"
    try:
        m2 = ConvolutionBlock(128, 128, 3, 1).eval().set_precision("bf16")
        x2 = torch.randn(32, 128, 80, 80, device="cuda")
        _repeat_equal(lambda: m2(x2))
    finally:
        os.environ.pop("SKY_HALO_NF8", None)


def test_attention_cores_are_deterministic_at_large_grids():
    import skyeye.core.models as M
    torch.manual_seed(6)
    t = M.TransformerLayer(256, 8).eval().set_precision("bf16")
    x = torch.randn(16, 256, 40, 40, device="cuda")
    a = _repeat_equal(lambda: t(x))
    one = t(x[15:])
    assert torch.equal(one[0], a[15]), "TransformerLayer: batch entry differs from the entry run alone"
    wsa = M.WindowedSelfAttention(128, 8, 4).eval().set_precision("bf16")
    xw = torch.randn(4096, 64, 128, device="cuda")            # 4096 windows of 8 x 8 tokens
    b = _repeat_equal(lambda: wsa(xw))
    assert torch.equal(wsa(xw[:7])[3], b[3])


@pytest.mark.parametrize("cin,cout,k,stride,hw,batch", [(128, 128, 3, 1, 80, 16), (64, 64, 3, 1, 160, 8), (128, 256, 3, 2, 80, 16), (256, 256, 1, 1, 80, 16)],
                         ids=["128to128_k3", "64to64_k3", "128to256_k3s2", "256to256_k1"])
def test_fp32_engine_is_deterministic_and_matches_torch(cin, cout, k, stride, hw, batch):
    """The exact (fp32 MFMA) instantiations of the same kernels under the same load: bit-identical runs, 1e-4 of the output range
    against torch's fp32 convolution (summation order only)."""
    torch.manual_seed(cin + cout + k)
    m = ConvolutionBlock(cin, cout, k, stride).eval().set_precision("fp32")
    x = torch.randn(batch, cin, hw, hw, device="cuda")
    got = _repeat_equal(lambda: m(x))
    w = m.conv.weight.detach().float().cuda() / np.float32(np.sqrt(1.0 + 1e-5))
    torch.backends.cudnn.allow_tf32 = False
    ref = F.silu(F.conv2d(x, w, stride=stride, padding=k // 2))
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
