"""GPU parity, whole graph: SkyEyeDetector.forward through the C ABI against the reference fixtures
(exact fp32 engine) and the bf16 production engine against the same fixtures (agreement rates)."""
import os

import numpy as np
import pytest
import torch

from cases import DETECTOR_CASES, HA_CASES, MODELS, variant_of
from helpers import build_detector, detector_params, variant_cfg, variant_enhanced
from parity import close, det_close, level_scales
from seeded import seeded_scene

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
_FULL = np.load(os.path.join(G, "detectors_full.npz"))
_HA = np.load(os.path.join(G, "detectors_ha.npz"))       # D5 wiring (head attention), tests/golden/make_golden.py ha


class _Full:
    def __getitem__(self, k):
        return _HA[k] if k in _HA.files else _FULL[k]


DET_FULL = _Full()
_SAMPLED = np.load(os.path.join(G, "detectors_sampled.npz"))


class _Sampled:
    def __getitem__(self, k):
        return _HA[k] if k in _HA.files else _SAMPLED[k]


DET_SAMPLED = _Sampled()

CASES = list(DETECTOR_CASES) + list(HA_CASES)
_MODELS = {}


def model_for(case):
    v = variant_of(case)
    if v not in _MODELS:
        m = build_detector(variant_cfg(v), variant_enhanced(v))
        m.load_state_dict({k: torch.from_numpy(np.asarray(a)) for k, a in detector_params(v).items()}, strict=True)
        _MODELS[v] = m.eval()
    return _MODELS[v]


def run(case, precision, as_uint8=False):
    m = model_for(case).set_precision(precision)
    h, w = case["hw"]
    frames = seeded_scene(case["batch"], h, w, case["seed"])
    # validate.py:236-238: img.float() / 255 on the CPU path (true division; torch's GPU kernel multiplies by 1/255)
    x = torch.from_numpy(frames if as_uint8 else frames.astype(np.float32) / np.float32(255.0)).cuda()
    det, raw = m(x)
    torch.cuda.synchronize()
    return det.cpu().numpy(), [r.cpu().numpy() for r in raw]


def tol_for(case):
    # Column tolerance (tests/parity.det_close) -- 1e-4, the north star's, for every graph but the Enhanced detector's.  Error budget of
    # round 4 (profiles/r04_f64_error_budget.json, tests/test_f64_error_budget.py: the graph evaluated in float64 on the CPU):
    #     enh_s_128x96, worst |d| / (1e-4 x scale):  reference fixture vs f64 0.587   fp32 engine vs f64 1.125   engine vs fixture 1.242
    # The reference's OWN fp32 result (oneDNN) is 0.59e-4 from the true value of its graph and the engine 1.13e-4 (the cross-layer
    # attention's column softmax over image rows, x4, turns the fp32 rounding of the K = 256 .. 512 projections into relative errors
    # of whole columns); two such implementations can differ by the sum, 1.71e-4.  Limit: 2e-4 (3e-4 in rounds 1 - 3).
    return 2e-4 if case.get("enhanced") else 1e-4


def iou_tol_for(case):
    # IoU >= 1 - 1e-4 on matched boxes -- now for EVERY plain graph.  Rounds 1 - 3 relaxed l_640 (1.56e-4 measured),
    # l_1280 (1.04e-4) and enh_s_128x96 (3.03e-4) to twice their measured values without knowing whose rounding it was.  The float64
    # budget (1 - min IoU against the f64 evaluation of the same graph):
    #     case            reference fixture   fp32 engine r03   fp32 engine r04 (two-level summation in the 3x3 kernels)
    #     l_640           0.76e-4             1.04e-4           0.79e-4   -> engine vs fixture 1.56e-4 -> 0.96e-4
    #     l_1280          0.75e-4             0.93e-4           0.81e-4   -> engine vs fixture 1.04e-4 -> 0.92e-4
    #     s_1280          0.33e-4             0.52e-4           0.48e-4   -> engine vs fixture 0.60e-4 -> 0.50e-4   (control)
    #     enh_s_128x96    1.66e-4             1.86e-4           2.00e-4   -> engine vs fixture 3.03e-4 -> 3.04e-4
    # The engine's share WAS the larger one on the skyeye_l graphs: a single fp32 accumulator chain over K = 9 x Cin made the 3x3 layers
    # 2.4 - 4.4 x less exact than oneDNN's (tools/fp32_layer_error.py); with partial sums every three taps they are as exact
    # (1.4 - 1.6e-7 relative rms against 1.6 - 1.7e-7) and both cases pass at 1e-4.  The Enhanced graph's fixture is itself 1.66e-4 from
    # the float64 result -- no fp32 implementation can be asked to sit closer to it than that --: its limit is the sum of the two
    # measured distances from float64, 1.66e-4 + 2.00e-4 (6.1e-4 = "twice what we saw" in rounds 1 - 3).
    return {"enh_s_128x96": 3.7e-4}.get(case["name"], 1e-4)


def check_against_fixture(case, det, raw, tol):
    name = case["name"]
    scales = level_scales(case["hw"])
    if case["store"] == "full":
        det_close(det, DET_FULL[f"{name}.det"], scales, tol, iou_tol_for(case))
        for i, r in enumerate(raw):
            close(r, DET_FULL[f"{name}.raw{i}"], rtol=5e-5 * tol / 1e-4)
    else:
        flat = det.reshape(-1, det.shape[-1])
        rows = DET_SAMPLED[f"{name}.rows"]
        det_close(flat[rows], DET_SAMPLED[f"{name}.det_rows"], np.tile(scales, (case["batch"], 1))[rows], tol, iou_tol_for(case))
        for i, r in enumerate(raw):
            rf = r.reshape(-1, r.shape[-1])
            close(rf[DET_SAMPLED[f"{name}.raw{i}_rows"]], DET_SAMPLED[f"{name}.raw{i}_vals"], rtol=5e-5 * tol / 1e-4)
        np.testing.assert_allclose(np.abs(flat.astype(np.float64)).mean(0), DET_SAMPLED[f"{name}.absmean"], rtol=1e-4)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_detector_fp32_matches_reference_fixture(case):
    det, raw = run(case, "fp32")
    assert det.shape[1] == sum(3 * (case["hw"][0] // s) * (case["hw"][1] // s) for s in (8, 16, 32))
    check_against_fixture(case, det, raw, tol_for(case))


def test_uint8_input_equals_float_input():
    case = CASES[0]
    d0, r0 = run(case, "fp32")
    d1, r1 = run(case, "fp32", as_uint8=True)
    assert np.array_equal(d0, d1)


@pytest.mark.parametrize("case", [c for c in CASES if c["store"] == "full"], ids=[c["name"] for c in CASES if c["store"] == "full"])
def test_detector_bf16_agreement(case):
    det, raw = run(case, "bf16")
    ref = DET_FULL[f"{case['name']}.det"]
    # logits: bf16 activations through ~100 layers; judged relative to the logit range
    for i, r in enumerate(raw):
        rr = DET_FULL[f"{case['name']}.raw{i}"]
        err = np.abs(r - rr).max() / max(1.0, np.abs(rr).max())
        # a MAX statistic over every logit of a random-weight network: 0.11 - 0.16 across equivalent roundings of the bf16 weights / sums
        # (round 4: exp2-domain weights, bias as the accumulators' initial value; the Enhanced graph's column softmax measured 0.160)
        assert err < (0.22 if case.get("enhanced") else 0.15), f"level {i}: bf16 logit error {err:.3f} of range"
    agree = (det[..., 5:].argmax(-1) == ref[..., 5:].argmax(-1)).mean()
    obj_err = np.abs(det[..., 4] - ref[..., 4]).max()
    print(f"{case['name']}: class agreement {agree:.4f}, max |d obj| {obj_err:.4f}")
    assert agree > 0.9
    # the Enhanced detector's column softmax (x4) amplifies bf16 rounding of the projections
    assert obj_err < (0.35 if case.get("enhanced") else 0.15)


_BIG = [c for c in CASES if c["store"] == "sampled"]


@pytest.mark.parametrize("case", _BIG, ids=[c["name"] for c in _BIG])
def test_detector_bf16_against_sampled_reference_fixture(case):
    """The bf16 engine at the full sizes (640, 1280; skyeye_s, skyeye_l, attention heads) against the reference's own rows:
    class argmax, objectness, IoU of the decoded boxes of the sampled rows.  Floors sit under the measured values (printed)."""
    from parity import record_agreement, row_iou
    det, raw = run(case, "bf16")
    name = case["name"]
    flat = det.reshape(-1, det.shape[-1])
    rows = DET_SAMPLED[f"{name}.rows"]
    ref = DET_SAMPLED[f"{name}.det_rows"]
    got = flat[rows]
    conf = ref[:, 4] > 0.25
    cls = float((got[conf, 5:].argmax(-1) == ref[conf, 5:].argmax(-1)).mean())
    ri = row_iou(got, ref, conf)
    dobj = float(np.abs(got[:, 4] - ref[:, 4]).mean())
    logit_err = max(float(np.abs(r.reshape(-1, r.shape[-1])[DET_SAMPLED[f"{name}.raw{i}_rows"]] - DET_SAMPLED[f"{name}.raw{i}_vals"]).max()
                          / max(1.0, np.abs(DET_SAMPLED[f"{name}.raw{i}_vals"]).max())) for i, r in enumerate(raw))
    record_agreement(f"{name} bf16 vs reference rows", cls_agree=cls, row_iou_mean=float(ri.mean()), row_iou_gt50=float((ri > 0.5).mean()),
                     mean_dobj=dobj, logit_err_of_range=logit_err, confident_rows=int(conf.sum()))
    assert np.isfinite(det).all()
    # floors per case: 0.97 x the measured class agreement / row IoU (round 3, profiles/r03_parity_margins.json), objectness and logit
    # errors 1.5 x measured
    lo = {"s_640": (0.955, 0.92, 0.964, 0.0056, 0.056), "s_1280": (0.956, 0.925, 0.965, 0.0047, 0.057), "l_640": (0.942, 0.838, 0.921, 0.0083, 0.072),
          "l_1280": (0.946, 0.852, 0.933, 0.0075, 0.091), "ha_s_1280": (0.96, 0.935, 0.968, 0.0062, 0.047)}.get(name, (0.93, 0.8, 0.92, 0.01, 0.1))
    assert cls > lo[0] and float(ri.mean()) > lo[1] and float((ri > 0.5).mean()) > lo[2] and dobj < lo[3] and logit_err < lo[4], (cls, ri.mean(), dobj, logit_err, lo)


def test_train_mode_returns_raw_only():
    case = CASES[0]
    m = model_for(case).set_precision("fp32")
    x = torch.from_numpy(seeded_scene(1, 64, 64, 3)).cuda().float() / 255.0
    m.train()
    out = m(x)
    m.eval()
    assert isinstance(out, list) and len(out) == 3 and out[0].shape[-1] == 15
