"""A cached engine must notice every weight change, also the ones that bypass the module that owns it:
load_state_dict on a plain nn.Module child, on the wrapper's inner module, an in-place edit, an optimizer step.
CPU test of the fingerprint that `_engine()` compares (skyeye/core/models/_base.py); no compute."""
import torch

from helpers import build_detector, variant_cfg


def _model():
    return build_detector(variant_cfg("skyeye_s")).eval()


def test_fingerprint_is_stable_without_changes():
    m = _model()
    assert m._weights_fingerprint() == m._weights_fingerprint()


def test_child_load_state_dict_is_seen():
    m = _model()
    fp = m._weights_fingerprint()
    sd = {k: v + 1.0 if v.dtype.is_floating_point else v for k, v in m.backbone.state_dict().items()}
    m.backbone.load_state_dict(sd)                       # SkyEyeBackbone is a plain nn.Module
    assert m._weights_fingerprint() != fp


def test_inner_module_load_state_dict_is_seen():
    m = _model()
    fp = m._weights_fingerprint()
    inner = m.backbone.backbone
    inner.load_state_dict({k: v.clone() for k, v in inner.state_dict().items()})
    assert m._weights_fingerprint() != fp


def test_inplace_edit_and_optimizer_step_are_seen():
    m = _model()
    fp = m._weights_fingerprint()
    p = next(m.neck.parameters())
    with torch.no_grad():
        p.mul_(2.0)
    fp2 = m._weights_fingerprint()
    assert fp2 != fp
    opt = torch.optim.SGD([p], lr=0.1)
    p.grad = torch.ones_like(p)
    opt.step()
    assert m._weights_fingerprint() != fp2


def test_rebound_parameter_and_cast_are_seen():
    m = _model()
    fp = m._weights_fingerprint()
    layer = m.detection_head.detection_layers[0]
    layer.bias = torch.nn.Parameter(torch.zeros_like(layer.bias))
    fp2 = m._weights_fingerprint()
    assert fp2 != fp
    m.backbone.double()
    assert m._weights_fingerprint() != fp2


def test_refresh_weights_forces_a_repack():
    """`.data` edits are the one form no cheap signal covers (`.data` has its own version counter by design):
    they need refresh_weights(), as _base.py documents."""
    m = _model()
    fp = m._weights_fingerprint()
    next(m.neck.parameters()).data.mul_(2.0)
    assert m._weights_fingerprint() == fp
    m.refresh_weights()
    assert m._weights_fingerprint() != fp
