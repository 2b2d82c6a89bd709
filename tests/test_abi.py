"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol the header declares,
and the host logic (module tree <-> engine parameter spec, error paths) behaves -- no compute, no GPU."""
import ctypes
import os
import re

import pytest
import torch

from cases import BLOCK_CASES, MODELS
from helpers import build_detector, build_module

from skyeye import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "skyeye_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sky_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = N.lib()
    declared = header_symbols()
    assert set(declared) == set(N.SYMBOLS), "python binding and header disagree on the ABI surface"
    for name in declared:
        assert hasattr(L, name), f"{name} is declared in include/skyeye_hip.h but not exported"
    assert L.sky_abi_version() == 1


def test_struct_layout_is_checked():
    cfg = N.make_config("CONV_BLOCK", c_in=32, c_out=32, kernel_size=1, stride=1, activation=1)
    cfg.struct_size = 12
    h = ctypes.c_void_p()
    assert N.lib().sky_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"size mismatch" in N.lib().sky_last_error(None)


def _state_spec(m):
    return {k: tuple(v.shape) for k, v in m.state_dict().items()
            if v.dtype != torch.long}


IMPLEMENTED = {"ConvolutionBlock", "FocusBlock", "BottleneckBlock", "CSPBlock", "SPPBlock", "ChannelAttention",
               "SpatialAttention", "CombinedAttention", "Backbone", "FeatureNeck", "DetectionHead", "CrossLayerAttention",
               "CrossLayerAttentionD4", "TransformerLayer", "WindowedSelfAttention"}


@pytest.mark.parametrize("case", [c for c in BLOCK_CASES if c["kind"] in IMPLEMENTED], ids=lambda c: c["name"])
def test_engine_param_spec_equals_module_state_dict(case):
    m = build_module(case)
    assert dict(m.expected_state()) == _state_spec(m)


def test_enhanced_detector_param_spec_equals_state_dict():
    m = build_detector(MODELS["skyeye_s"], enhanced=True)
    spec = dict(m.expected_state())
    assert spec == _state_spec(m)
    assert spec["cross_attention_p5_p4.key_projection.weight"] == (256, 512, 1, 1)      # D4: key -> query width
    assert spec["cross_attention_p4_p3.output_projection.weight"] == (128, 128, 1, 1)


@pytest.mark.parametrize("model", sorted(MODELS))
def test_detector_param_spec_equals_state_dict(model):
    m = build_detector(MODELS[model])
    spec = dict(m.expected_state())
    assert spec == _state_spec(m)
    # names follow the reference's module tree
    assert "backbone.backbone.stage1.0.conv.conv.weight" in spec
    assert "backbone.backbone.stage3.2.channel_attention.shared_mlp.0.weight" in spec
    assert spec["backbone.backbone.stage3.2.spatial_attention.conv.weight"] == (1, 2, 7, 7)
    assert "neck.lateral_conv5.conv.weight" in spec and "neck.pan_conv5.cv3.bn.running_var" in spec
    assert spec["detection_head.detection_layers.2.weight"][0] == 3 * (MODELS[model]["nc"] + 5)


def test_detector_api_surface():
    m = build_detector(MODELS["skyeye_s"])
    assert m.stride.tolist() == [8, 16, 32]                     # detector.py:291-295
    assert m.names == [str(i) for i in range(10)]               # detector.py:298
    assert m.neck.out_channels == [128, 256, 512]               # D1/D2
    assert hasattr(m, "backbone") and hasattr(m, "neck") and hasattr(m, "detection_head")
    assert len([k for k in m.state_dict()]) > 400
    from skyeye.core.detector import SkyEyeDetector as ReadmePath   # README.md:41
    assert ReadmePath is type(m)
    from skyeye.core.models import construct_model, parse_model
    cfg = parse_model(dict(nc=4))
    assert cfg["base_channels"] == 64 and cfg["nc"] == 4 and cfg["anchors"] is None   # detector.py:393-405
    assert construct_model(dict(nc=4, depth_multiple=0.33, width_multiple=0.25), num_classes=6).cfg["nc"] == 6


def test_yaml_configs_load():
    from skyeye.core.models import SkyEyeDetector
    for name, (dm, wm) in dict(skyeye_s=(0.33, 0.5), skyeye_m=(0.67, 0.75), skyeye_l=(1.0, 1.0)).items():
        m = SkyEyeDetector(f"{name}.yaml")
        assert m.cfg["depth_multiple"] == dm and m.cfg["width_multiple"] == wm and m.cfg["nc"] == 10


def test_no_cpu_path():
    m = build_detector(MODELS["skyeye_s"])
    with pytest.raises(N.SkyEyeNativeError):
        m(torch.zeros(1, 3, 64, 64))
    if not torch.cuda.is_available():
        h = N.Handle(N.make_config("CONV_BLOCK", c_in=32, c_out=32, kernel_size=1, stride=1, activation=1))
        b = N.SkyBuffer()
        b.ndim = 4
        for i, s in enumerate((1, 32, 8, 8)):
            b.shape[i] = s
        with pytest.raises(N.SkyEyeNativeError, match="no HIP device|missing weight"):
            h.plan([b])


def test_unsupported_configurations_fail_loudly():
    from skyeye.core.models import ConvolutionBlock
    with pytest.raises(N.SkyEyeNativeError, match="multiple of"):
        ConvolutionBlock(30, 32, 1, 1).expected_state()
    with pytest.raises(N.SkyEyeNativeError, match="kernel_size"):
        ConvolutionBlock(32, 32, 5, 1).expected_state()
