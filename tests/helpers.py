"""Shared by the CPU and GPU tests: build the drop-in module for a golden case and its seeded state dict."""
import numpy as np

from seeded import seeded_input, seeded_tensor

import skyeye.core.models as M


def build_module(case):
    kind, a = case["kind"], dict(case["args"])
    if kind == "CrossLayerAttentionD4":
        return M.CrossLayerAttention(**a)        # key_channels != query_channels selects the D4 projections
    return getattr(M, kind)(**a) if hasattr(M, kind) else getattr(__import__("skyeye.core.models.detector", fromlist=[kind]), kind)(**a)


def seeded_state_for(module, seed):
    out = {}
    for k, v in module.state_dict().items():
        if k.rsplit(".", 1)[-1] == "relative_position_index":
            out[k] = v.numpy()
        else:
            out[k] = seeded_tensor(k, tuple(v.shape), seed)
    return out


def block_params(case):
    return seeded_state_for(build_module(case), case["seed"])


def block_inputs(case):
    return {k: seeded_input(case["name"] + "." + k, shp, case["seed"], lo, hi) for k, (shp, lo, hi) in case["inputs"].items()}


def build_detector(cfg, enhanced=False):
    cls = M.EnhancedSkyEyeDetector if enhanced else M.SkyEyeDetector
    return cls(dict(cfg))


_CALIB = None


def detector_params(variant):
    """Seeded weights of a detector variant ("skyeye_s", ..., "skyeye_s_enh") with the calibrated BatchNorm
    running statistics of tests/golden/bn_calib.npz (see cases.WSEED)."""
    import os
    from cases import MODELS, WSEED
    global _CALIB
    if _CALIB is None:
        g = os.path.join(os.path.dirname(__file__), "golden")
        _CALIB = {}
        for f in ("bn_calib.npz", "bn_calib_ha.npz"):
            z = np.load(os.path.join(g, f))
            _CALIB.update({k: z[k] for k in z.files})
    P = seeded_state_for(build_detector(variant_cfg(variant), variant_enhanced(variant)), WSEED[variant])
    for k in list(P):
        ck = f"{variant}:{k}"
        if ck in _CALIB:
            P[k] = _CALIB[ck]
    return P


def variant_enhanced(variant):
    base = variant[:-3] if variant.endswith("_ha") else variant
    return base.endswith("_enh")


def variant_cfg(variant):
    """cases.MODELS entry of a detector variant name ("skyeye_s", "skyeye_s_enh", "skyeye_s_ha", ...)."""
    from cases import MODELS
    ha = variant.endswith("_ha")
    base = variant[:-3] if ha else variant
    if base.endswith("_enh"):
        base = base[:-4]
    cfg = dict(MODELS[base])
    if ha:
        cfg["head_attention"] = True
    return cfg


def load_seeded(module, seed):
    import torch
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in seeded_state_for(module, seed).items()}
    module.load_state_dict(sd, strict=True)
    return module.eval()


