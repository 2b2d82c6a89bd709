"""Export of the engine's packed weight file: BatchNorm folding and the (ky, kx, cin) K order of
sky_packed_read checked against numpy on seeded parameters (the reference's fuse_conv_and_bn does not exist; its
fused_forward, blocks.py:39-41, defines the folded form: conv with w * gamma / sqrt(var + eps), bias beta - mean * scale)."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import load_seeded, seeded_state_for

import skyeye.core.models as M

pytestmark = pytest.mark.gpu


def _bf16_bits(a):
    """float32 -> bf16 bits, round to nearest even (what the engine's packer does)."""
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_export_folds_batchnorm_and_orders_k(tmp_path, prec):
    m = load_seeded(M.ConvolutionBlock(16, 48, 3, 1), 33).set_precision(prec)
    P = seeded_state_for(m, 33)
    x = torch.zeros(1, 16, 16, 16, device="cuda")
    path = os.path.join(tmp_path, "conv.npz")
    index = m.export_engine(path, x)
    z = np.load(path)
    assert json.loads(bytes(z["index"]).decode()) == index and len(index) == 1
    e = index[0]
    assert e["name"] == "conv.weight" and e["cout"] == 48 and e["kernel_size"] == 3 and e["cin"] == 16
    w, b = z["conv0.weight"], z["conv0.bias"]
    scale = (P["bn.weight"] / np.sqrt(P["bn.running_var"] + np.float32(1e-5))).astype(np.float32)
    want_b = (P["bn.bias"] - P["bn.running_mean"] * scale).astype(np.float32)
    want_w = (P["conv.weight"] * scale[:, None, None, None]).transpose(0, 2, 3, 1).reshape(48, -1)       # [cout][(ky, kx, cin)]
    K = want_w.shape[1]
    assert w.shape[0] >= 48 and w.shape[1] >= K
    sc = z["conv0.scale"]
    if prec == "bf16":
        # a SiLU layer of the bf16 engine lives in the exp2 domain: weights and bias are stored times log2 e, scale = ln 2
        np.testing.assert_allclose(sc[:48], np.log(2.0), rtol=1e-6)
        want_b = (want_b.astype(np.float64) * np.log2(np.e)).astype(np.float32)
        want_w = (want_w.astype(np.float64) * np.log2(np.e)).astype(np.float32)
    else:
        assert (sc == 1.0).all()
    np.testing.assert_allclose(b[:48], want_b, rtol=1e-6, atol=1e-7)
    if prec == "fp32":
        np.testing.assert_allclose(w[:48, :K], want_w, rtol=1e-6, atol=1e-8)
        assert not w[48:].any() and not w[:, K:].any()                       # padding rows / K tail are zeros
    else:
        assert w.dtype == np.uint16
        got, ref = w[:48, :K].astype(np.int32), _bf16_bits(want_w).astype(np.int32)
        assert np.abs(got - ref).max() <= 1                                   # one bf16 ulp where the fp32 product differs in the last bit


def test_export_detector_lists_every_convolution(tmp_path):
    from cases import MODELS
    from helpers import build_detector, detector_params
    m = build_detector(MODELS["skyeye_s"])
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in detector_params("skyeye_s").items()}, strict=True)
    m.eval().set_precision("bf16")
    index = m.export_engine(os.path.join(tmp_path, "s.npz"), torch.zeros(1, 3, 64, 64, dtype=torch.uint8, device="cuda"))
    names = [e["name"] for e in index]
    assert 60 < len(index) <= 78                                     # 75 ConvolutionBlocks + 3 detection levels, cv1|cv2 pairs packed as one
    assert any(n.startswith("detection_head.detection_layers.0") for n in names)
    assert any(n.endswith("cv1.conv.weight") for n in names)
