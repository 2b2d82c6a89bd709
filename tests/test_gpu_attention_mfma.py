"""GPU parity of the MFMA flash attention core (csrc/k_attn.hip: attention_mfma_kernel) inside TransformerLayer
(reference attention.py:244-309) for the bf16 engine: against the CPU oracle and against the fp32-VALU core of the same
engine (SKY_ATTN_VALU=1) on the same bf16 inputs."""
import os

import numpy as np
import pytest
import torch

from helpers import load_seeded, seeded_state_for
from seeded import seeded_input

import skyeye.core.models as M

pytestmark = pytest.mark.gpu

# (dim, heads, B, H, W): head dims 32 / 64 / 128; token counts that are / are not multiples of the 64-key block
CASES = [(128, 4, 2, 8, 12), (256, 4, 2, 16, 16), (256, 2, 1, 10, 10), (512, 8, 1, 20, 20), (64, 2, 3, 8, 8)]


def _oracle():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import skyeye_oracle as O
    return O


def _run(m, x, valu):
    if valu:
        os.environ["SKY_ATTN_VALU"] = "1"
    try:
        y = m(x)
        torch.cuda.synchronize()
        return y.cpu().numpy()
    finally:
        os.environ.pop("SKY_ATTN_VALU", None)


@pytest.mark.parametrize("case", CASES, ids=["d%d_h%d_b%d_%dx%d" % c for c in CASES])
def test_mfma_attention_matches_oracle_and_valu_core(case):
    dim, heads, B, H, W = case
    O = _oracle()
    m = load_seeded(M.TransformerLayer(dim, heads), 61).set_precision("bf16")
    P = seeded_state_for(m, 61)
    x = seeded_input("attn.x.%d.%d" % (dim, H), (B, dim, H, W), 13, -1.0, 1.0)
    ref = O.transformer_layer(P, "", x, heads)
    xg = torch.from_numpy(x).cuda()
    y_mfma = _run(m, xg, False)
    y_valu = _run(m, xg, True)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.isfinite(y_mfma).all()
    # bf16 engine vs fp32 oracle: the same budget the other bf16 block tests use
    assert float(np.abs(y_mfma - ref).max()) <= 4e-2 * scale
    # the two cores see identical bf16 q, k, v; the MFMA core additionally rounds P to bf16 before P.V
    assert float(np.abs(y_mfma - y_valu).max()) <= 2e-2 * scale


# WindowedSelfAttention (reference attention.py:312-399) with 8 x 8 windows = 64 tokens: relative position bias, optional mask
WSA_CASES = [(64, 2, 6, False), (128, 4, 8, True), (256, 4, 4, True), (128, 1, 3, False)]     # (dim, heads, B_, masked)


@pytest.mark.parametrize("case", WSA_CASES, ids=["d%d_h%d_n%d_m%d" % c for c in WSA_CASES])
def test_mfma_windowed_attention_matches_oracle_and_valu_core(case):
    dim, heads, Bw, masked = case
    O = _oracle()
    m = load_seeded(M.WindowedSelfAttention(dim, 8, heads), 62).set_precision("bf16")
    P = seeded_state_for(m, 62)
    x = seeded_input("wsa.x.%d.%d" % (dim, Bw), (Bw, 64, dim), 14, -1.0, 1.0)
    mask = None
    if masked:
        nW = 2 if Bw % 2 == 0 else 1
        r = np.random.default_rng(3)
        mask = np.where(r.uniform(0, 1, (nW, 64, 64)) < 0.2, -100.0, 0.0).astype(np.float32)     # Swin-style additive mask
    ref = O.windowed_self_attention(P, "", x, 8, heads, mask)
    xg = torch.from_numpy(x).cuda()
    mg = None if mask is None else torch.from_numpy(mask).cuda()

    def run(valu):
        if valu:
            os.environ["SKY_ATTN_VALU"] = "1"
        try:
            y = m(xg, mg)
            torch.cuda.synchronize()
            return y.cpu().numpy()
        finally:
            os.environ.pop("SKY_ATTN_VALU", None)

    y_mfma, y_valu = run(False), run(True)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.isfinite(y_mfma).all()
    assert float(np.abs(y_mfma - ref).max()) <= 4e-2 * scale
    assert float(np.abs(y_mfma - y_valu).max()) <= 2e-2 * scale


# 8 x 8 windows with 32-channel heads, heads % 4 == 0, no mask: window_attention_kernel (one wave per (window, head), bias in
# registers, V through a transposed LDS read).  Same arithmetic in the same order as the general flash kernel: bit-identical.
WIN_CASES = [(128, 4, 8), (256, 8, 5), (128, 4, 1), (512, 16, 3)]     # (dim, heads, B_)


@pytest.mark.parametrize("case", WIN_CASES, ids=["d%d_h%d_n%d" % c for c in WIN_CASES])
def test_window_kernel_matches_oracle_and_is_bit_identical_to_flash_kernel(case):
    dim, heads, Bw = case
    O = _oracle()
    m = load_seeded(M.WindowedSelfAttention(dim, 8, heads), 63).set_precision("bf16")
    P = seeded_state_for(m, 63)
    x = seeded_input("win.x.%d.%d" % (dim, Bw), (Bw, 64, dim), 15, -1.0, 1.0)
    ref = O.windowed_self_attention(P, "", x, 8, heads, None)
    xg = torch.from_numpy(x).cuda()

    def run(switch):
        if switch:
            os.environ[switch] = "1"
        try:
            y = m(xg)
            torch.cuda.synchronize()
            return y.cpu().numpy()
        finally:
            if switch:
                os.environ.pop(switch, None)

    y_win, y_flash, y_valu = run(None), run("SKY_NO_WINATTN"), run("SKY_ATTN_VALU")
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.isfinite(y_win).all()
    assert float(np.abs(y_win - ref).max()) <= 4e-2 * scale
    assert float(np.abs(y_win - y_valu).max()) <= 2e-2 * scale
    assert np.array_equal(y_win, y_flash), "window kernel differs from the general flash kernel"


def test_window_kernel_on_a_feature_map_is_bit_identical_to_flash_kernel():
    """The detector's wiring: windows addressed in place on a [B, H, W, C] map (config 3, head_attention; 4 and 8 heads)."""
    from helpers import build_detector, detector_params, variant_cfg
    from seeded import seeded_scene
    m = build_detector(variant_cfg("skyeye_s_ha"))
    m.load_state_dict({k: torch.from_numpy(np.asarray(a)) for k, a in detector_params("skyeye_s_ha").items()}, strict=True)
    m.eval().set_precision("bf16")
    x = torch.from_numpy(seeded_scene(3, 256, 384, 80)).cuda()

    def run(switch):
        if switch:
            os.environ[switch] = "1"
        try:
            det, raw = m(x)
            torch.cuda.synchronize()
            return [det.cpu().numpy()] + [r.cpu().numpy() for r in raw]
        finally:
            if switch:
                os.environ.pop(switch, None)

    a, b = run(None), run("SKY_NO_WINATTN")
    for u, v in zip(a, b):
        assert np.isfinite(u).all() and np.array_equal(u, v)
