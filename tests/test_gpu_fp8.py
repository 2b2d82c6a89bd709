"""The fp8 engine (SKY_FP8: OCP e4m3fn weights with one scale per output channel, activations with one calibrated scale per
tensor, bf16 stem, fp32 accumulation; BASELINE.json configs[4]).

  * the host quantizer and every convolution kernel family against a numpy / torch emulation of exactly that arithmetic
    (quantise inputs and weights, convolve the dequantised values in fp32, quantise the output): equal up to the last e4m3 step;
  * whole detectors against the reference fixtures and the fp32 engine: class-argmax agreement, IoU-matched post-NMS box
    agreement and mAP, with the rates PRINTED (tests/conftest.py) -- 1e-4 is not a meaningful gate at 3 mantissa bits;
  * 1536 x 1536 (config 5's size, 145 152 rows): determinism, batch independence, finiteness."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import skyeye.core.models as M
from helpers import build_detector, detector_params, load_seeded, variant_cfg
from parity import box_agreement, e4m3_table, quantize_e4m3, record_agreement, row_iou
from seeded import seeded_input, seeded_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _detector(variant, precision):
    m = build_detector(variant_cfg(variant))
    m.load_state_dict({k: torch.from_numpy(np.asarray(a)) for k, a in detector_params(variant).items()}, strict=True)
    return m.eval().set_precision(precision)


def test_quantize_helper_is_the_e4m3_grid():
    t = e4m3_table()
    fin = t[np.isfinite(t)]
    assert np.array_equal(quantize_e4m3(fin), fin)
    assert quantize_e4m3(np.float32([1000.0, -1e9, 0.0009, 464.0]))[:3].tolist() == [448.0, -448.0, 0.0]
    assert quantize_e4m3(np.float32([17.0]))[0] == 16.0 and quantize_e4m3(np.float32([19.0]))[0] == 20.0      # ties to even


# (cin, cout, k, stride, B, H, W): one case per kernel family the fp8 dispatch reaches
CONV_CASES = [
    (128, 128, 3, 1, 2, 48, 48),     # halo tile kernel, one 128-byte chunk, 128-channel tiles (16x16x128 MFMA)
    (256, 64, 3, 1, 2, 32, 32),      # halo tile kernel, two chunks, 64-channel tiles
    (64, 64, 3, 1, 2, 48, 64),       # narrow halo kernel, 64 bytes per pixel
    (32, 32, 3, 1, 2, 64, 64),       # narrow halo kernel, 32 bytes per pixel
    (128, 256, 3, 2, 2, 64, 64),     # halo tile kernel, stride 2
    (64, 128, 3, 2, 2, 64, 64),      # streaming kernel, 3x3 stride 2
    (32, 64, 3, 2, 2, 64, 64),       # tile kernel (implicit GEMM)
    (64, 64, 1, 1, 2, 40, 40),       # streaming kernel, one K-step
    (128, 256, 1, 1, 2, 40, 40),     # streaming kernel, two N tiles
    (512, 256, 1, 1, 2, 20, 20),     # streaming kernel, weight ring / large K
    (256, 48, 1, 1, 2, 24, 24),      # tile kernel (Cout not a multiple of 32)
]


def _emulate_conv(h, x, k, stride, act=True):
    """The fp8 arithmetic restated: e4m3(x / s_in) * s_in conv e4m3(w / s_w) * s_w + b -> SiLU -> e4m3(. / s_out) * s_out,
    with the engine's own packed weights and scales (test_packed_weights_* checks those against numpy)."""
    pw = h.packed_weights()[0]
    sc = h.scales()
    s_in, s_out = float(sc[0]), float(sc[1])
    cout, cin = pw["cout"], pw["cin"]
    tab = e4m3_table()
    w = tab[pw["weight"][:cout, :k * k * cin]] * pw["scale"][:cout, None]
    w = w.reshape(cout, k, k, cin).transpose(0, 3, 1, 2)
    inv_in, inv_out = np.float32(1.0) / np.float32(s_in), np.float32(1.0) / np.float32(s_out)     # the kernels multiply by 1 / scale
    xq = quantize_e4m3(x.astype(np.float32) * inv_in) * np.float32(s_in)
    y = F.conv2d(torch.from_numpy(xq), torch.from_numpy(np.ascontiguousarray(w)), torch.from_numpy(pw["bias"][:cout]), stride=stride, padding=k // 2)
    if act:
        y = F.silu(y)
    return quantize_e4m3(y.numpy() * inv_out) * np.float32(s_out), s_out


@pytest.mark.parametrize("case", CONV_CASES, ids=["%d-%d_k%ds%d_b%d_%dx%d" % c for c in CONV_CASES])
def test_fp8_conv_kernels_match_emulation(case):
    cin, cout, k, stride, B, H, W = case
    m = load_seeded(M.ConvolutionBlock(cin, cout, k, stride), 31).set_precision("fp8")
    x = seeded_input("fp8.x.%d.%d" % (cin, H), (B, cin, H, W), 9, -3.0, 3.0)
    y = m(torch.from_numpy(x).cuda()).cpu().numpy()
    h = m._engine([torch.from_numpy(x).cuda()])
    ref, s_out = _emulate_conv(h, x, k, stride)
    assert np.isfinite(y).all()
    # output values sit on the e4m3 grid of the output scale
    assert np.array_equal(quantize_e4m3(y * (np.float32(1.0) / np.float32(s_out))) * np.float32(s_out), y)
    diff = np.abs(y - ref)
    step = np.maximum(np.abs(ref), s_out * 2.0 ** -6) * 0.126          # one e4m3 step is 1/8 of the binade (12.5 %)
    frac_exact = float((diff == 0).mean())
    record_agreement("conv %d->%d k%d s%d" % (cin, cout, k, stride), exact=frac_exact, max_steps=float((diff / step).max()))
    bad = np.argwhere(diff > step)
    detail = "; ".join(f"{tuple(int(v) for v in i)}: got {y[tuple(i)]:.6g} ref {ref[tuple(i)]:.6g}" for i in bad[:8])
    assert len(bad) == 0, f"{len(bad)} outputs more than one e4m3 step from the emulation (s_out {s_out:.4g}): {detail}"
    assert frac_exact > 0.97, frac_exact                                # fp32 summation order flips a few roundings


def test_fp8_bottleneck_residual_matches_emulation():
    """x + cv2(cv1(x)) in place on one fp8 buffer: the residual is read in the output's own scale."""
    m = load_seeded(M.BottleneckBlock(128, 128, shortcut=True, expansion=1.0), 7).set_precision("fp8")
    x = seeded_input("fp8.res", (2, 128, 40, 40), 3, -2.0, 2.0)
    xt = torch.from_numpy(x).cuda()
    y = m(xt).cpu().numpy()
    ref = load_seeded(M.BottleneckBlock(128, 128, shortcut=True, expansion=1.0), 7).set_precision("fp32")(xt).cpu().numpy()
    err = np.abs(y - ref).max() / np.abs(ref).max()
    corr = float(np.corrcoef(y.ravel(), ref.ravel())[0, 1])
    record_agreement("bottleneck 128 +res", max_err_of_range=err, corr=corr)
    assert err < 0.12 and corr > 0.995, (err, corr)


def test_packed_weights_match_numpy_quantizer():
    """Per-output-channel scale = max |BN-folded w| / 448; bytes = round-to-nearest-even e4m3 of w / scale (engine.cpp: f32_to_e4m3)."""
    m = load_seeded(M.ConvolutionBlock(64, 96, 3, 1), 13).set_precision("fp8")
    x = torch.from_numpy(seeded_input("fp8.q", (1, 64, 16, 16), 1, -1.0, 1.0)).cuda()
    m(x)
    pw = m._engine([x]).packed_weights()[0]
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    g = sd["bn.weight"] / np.sqrt(sd["bn.running_var"] + np.float32(1e-5))
    w = (sd["conv.weight"] * g[:, None, None, None]).astype(np.float32)                 # [cout, cin, ky, kx]
    w = w.transpose(0, 2, 3, 1).reshape(96, -1)                                          # K = (ky, kx, cin)
    scale = (np.abs(w).max(1) / np.float32(448.0)).astype(np.float32)
    np.testing.assert_allclose(pw["scale"][:96], scale, rtol=1e-6)
    want = quantize_e4m3(w / pw["scale"][:96, None])
    got = e4m3_table()[pw["weight"][:96, :w.shape[1]]]
    assert pw["weight"].dtype == np.uint8 and np.array_equal(got, want)
    assert not pw["weight"][:96, w.shape[1]:].any() and not pw["weight"][96:].any()      # K padding and row padding are zero


def test_fp8_needs_calibration_and_scales_round_trip():
    from skyeye import _native as N
    m = load_seeded(M.ConvolutionBlock(64, 64, 3, 1), 5)
    x = torch.from_numpy(seeded_input("fp8.cal", (2, 64, 24, 24), 2, -2.0, 2.0)).cuda()
    h = N.Handle(N.make_config("CONV_BLOCK", dtype=N.SKY_FP8, c_in=64, c_out=64, kernel_size=3, stride=1, activation=1))
    h.load_weights(m._named_weights())
    h.plan([N.buffer_from_tensor(x)])
    out = torch.empty(h.output_shapes()[0], device="cuda")
    with pytest.raises(N.SkyEyeNativeError, match="sky_calibrate"):
        h.forward([N.buffer_from_tensor(x)], [N.buffer_from_tensor(out)], 0)
    h.calibrate([N.buffer_from_tensor(x)], 0)
    h.forward([N.buffer_from_tensor(x)], [N.buffer_from_tensor(out)], 0)
    torch.cuda.synchronize()
    a = out.clone()
    sc = h.scales()
    assert sc.shape[0] >= 2 and float(sc[0]) == pytest.approx(float(np.abs(x.cpu().numpy()).max()) / 448.0, rel=0.01)
    h2 = N.Handle(N.make_config("CONV_BLOCK", dtype=N.SKY_FP8, c_in=64, c_out=64, kernel_size=3, stride=1, activation=1))
    h2.load_weights(m._named_weights())
    h2.plan([N.buffer_from_tensor(x)])
    h2.set_scales(sc)                                                   # restored scales == calibrated scales
    h2.forward([N.buffer_from_tensor(x)], [N.buffer_from_tensor(out)], 0)
    torch.cuda.synchronize()
    assert torch.equal(out, a)


def test_fp8_rejects_attention_graphs():
    from skyeye import _native as N
    m = load_seeded(M.TransformerLayer(64, 4), 3).set_precision("fp8")
    with pytest.raises(N.SkyEyeNativeError, match="fp8"):
        m(torch.zeros(1, 64, 8, 8).cuda())


# ---- whole detectors ------------------------------------------------------------------------------------------------------------
def _nms_np(det, conf=0.25):
    from skyeye.utils.metrics import non_max_suppression
    return [o.cpu().numpy() for o in non_max_suppression(det, conf, 0.45, mode="corrected")]


@pytest.mark.parametrize("variant,hw,batch", [("skyeye_s", (256, 256), 4), ("skyeye_l", (128, 128), 2), ("skyeye_l", (96, 160), 2), ("skyeye_l", (640, 640), 2),
                                              ("skyeye_s", (640, 640), 2), ("skyeye_s", (1280, 1280), 1)])
def test_fp8_detector_agreement_with_fp32_engine(variant, hw, batch):
    """fp8 vs the fp32 engine (itself pinned to the reference fixtures at these sizes, test_gpu_detector.py) on seeded scenes:
    class argmax on confident rows, objectness error, IoU-matched post-NMS agreement.  Rates are printed; the asserted floors
    sit well under the measured values (r02: see DESIGN.md section 4)."""
    sys.path.insert(0, ROOT)
    from bench import calibrate_objectness
    x = torch.from_numpy(seeded_scene(batch, hw[0], hw[1], 21)).cuda()
    ref_m = _detector(variant, "fp32")
    shift = calibrate_objectness(ref_m, x, 0.01, 0.25)
    ref, _ = ref_m(x)
    out = {}
    for prec in ("bf16", "fp8"):
        m = _detector(variant, prec)
        no = ref.shape[-1]
        with torch.no_grad():
            for layer in m.detection_head.detection_layers:
                layer.bias.view(-1, no)[:, 4] += shift
        m.refresh_weights()
        det, _ = m(x)
        assert bool(torch.isfinite(det).all())
        conf_rows = ref[..., 4] > 0.25
        cls_agree = float((det[..., 5:].argmax(-1) == ref[..., 5:].argmax(-1))[conf_rows].float().mean()) if int(conf_rows.sum()) else 1.0
        obj_err = float((det[..., 4] - ref[..., 4]).abs().max())
        obj_mean = float((det[..., 4] - ref[..., 4]).abs().mean())
        # the same rows (cell, anchor) of both engines: IoU of the decoded boxes where the reference is confident
        ri = row_iou(det.cpu().numpy().reshape(-1, det.shape[-1]), ref.cpu().numpy().reshape(-1, ref.shape[-1]), conf_rows.cpu().numpy().reshape(-1))
        ka, kb = _nms_np(det), _nms_np(ref)
        rates = [box_agreement(a, b, 0.5) for a, b in zip(ka, kb)]
        rates9 = [box_agreement(a, b, 0.9) for a, b in zip(ka, kb)]
        n_ref = sum(len(b) for b in kb)
        match50 = float(np.mean([r[0] for r in rates])); match90 = float(np.mean([r[0] for r in rates9])); miou = float(np.mean([r[1] for r in rates]))
        record_agreement(f"{variant} {hw[0]}x{hw[1]} {prec} vs fp32", cls_agree=cls_agree, row_iou_mean=float(ri.mean()), row_iou_gt90=float((ri > 0.9).mean()),
                         row_iou_gt50=float((ri > 0.5).mean()), max_dobj=obj_err, mean_dobj=obj_mean, boxes_ref=n_ref,
                         nms_matched_iou50=match50, nms_matched_iou90=match90, nms_mean_iou=miou)
        out[prec] = dict(cls=cls_agree, dobj=obj_mean, row_iou=float(ri.mean()), row90=float((ri > 0.9).mean()), row50=float((ri > 0.5).mean()), nms50=match50, n=n_ref)
    assert out["fp8"]["n"] > 10
    # Floors = 0.9 x the value measured for THIS case in round 3 (printed above per case; profiles/r03_parity_margins.json).  These
    # networks have RANDOM weights: nothing in them damps noise, so the per-layer rounding error (bf16 2^-9, e4m3 2^-4 relative) adds
    # up over the 75-107 layers.  tests/test_fp8_sensitivity.py (CPU emulation of the same arithmetic, layer by layer) shows what a
    # per-layer precision choice could buy: row IoU 0.44 (all fp8) -> 0.75 needs 24 of the 75 layers in bf16, class agreement 0.9
    # needs 48 -- the early, large layers first.  What the numbers here pin is that the engine computes the fp8 arithmetic it states
    # (test_fp8_conv_kernels_match_emulation: 99.95 % of outputs bit-equal to the emulation) and does not degrade.
    f, b = out["fp8"], out["bf16"]
    lo = FP8_ENGINE_FLOORS[(variant, hw)]
    assert f["cls"] >= lo[0] and f["row_iou"] >= lo[1] and f["row50"] >= lo[2] and f["dobj"] <= lo[3], (out, lo)
    assert b["cls"] > 0.85 and b["row_iou"] > 0.8 and b["row50"] > 0.95 and b["nms50"] > 0.78 and b["dobj"] < 0.002, out


FP8_ENGINE_FLOORS = {   # (class agreement, row IoU mean, rows with IoU > 0.5) x 0.8, mean |d obj| x 1.25 of the WORSE of the round-3 and round-4
    # measurements.  Round 4 changed the rounding realisation of the bf16 stem twice without changing its precision: the weights of every bf16
    # SiLU layer are kept times log2 e (exp2-domain activation) and the bias is the accumulators' initial value (a sum rounds as b + k0 + k1 ..
    # instead of k0 + k1 .. + b).  What such a change alone does to a reduced-precision engine on these RANDOM-weight networks was measured
    # (profiles/r04_pre_scale_eval.jsonl: the bf16 engine with the weights held times 1.0 / 1.2 / log2 e / 1.7 / 2.0 -- 2.0 is bit-identical
    # to 1.0): relative L2 error of the raw logits 0.0142 .. 0.0190 on skyeye_s, the threshold rates move by +- 10 - 15 % with it; the two small
    # skyeye_l cases hold 12 - 14 reference boxes.  0.9 x one measurement (rounds 2 - 3) was inside that spread: 0.8 x the worse of two is not.
    ("skyeye_s", (256, 256)): (0.633, 0.358, 0.334, 0.0071),
    ("skyeye_l", (128, 128)): (0.520, 0.403, 0.320, 0.0148),
    ("skyeye_l", (96, 160)): (0.578, 0.396, 0.376, 0.0140),
    ("skyeye_l", (640, 640)): (0.528, 0.253, 0.156, 0.0144),
    ("skyeye_s", (640, 640)): (0.657, 0.416, 0.437, 0.0086),
    ("skyeye_s", (1280, 1280)): (0.717, 0.455, 0.500, 0.0085),
}


def test_fp8_map_against_fp32_engine():
    """f2 accounting: detections of the fp8 engine scored against the fp32 engine's as ground truth."""
    sys.path.insert(0, ROOT)
    import skyeye.utils.metrics as SM
    from bench import calibrate_objectness
    x = torch.from_numpy(seeded_scene(4, 320, 320, seed=11)).cuda()
    ref_m = _detector("skyeye_s", "fp32")
    shift = calibrate_objectness(ref_m, x, 0.01, 0.25)
    outs = {"fp32": SM.non_max_suppression(ref_m(x)[0], 0.25, 0.45, mode="corrected")}
    for prec in ("bf16", "fp8"):
        m = _detector("skyeye_s", prec)
        with torch.no_grad():
            for layer in m.detection_head.detection_layers:
                layer.bias.view(-1, 15)[:, 4] += shift
        m.refresh_weights()
        outs[prec] = SM.non_max_suppression(m(x)[0], 0.25, 0.45, mode="corrected")
    labels = [torch.cat([o[:, 5:6], o[:, :4]], 1) for o in outs["fp32"]]
    r8 = SM.mean_average_precision(outs["fp8"], labels)
    r16 = SM.mean_average_precision(outs["bf16"], labels)
    record_agreement("skyeye_s 320 mAP vs fp32 engine", fp8_map50=r8["map50"], fp8_map=r8["map"], bf16_map50=r16["map50"], bf16_map=r16["map"])
    # floors: 0.8 x the worst measurement.  Round 3: fp8 mAP@.5 0.178, bf16 0.971 / 0.873; round 4 with the exp2-domain weights (another bf16
    # rounding of the same weights, see FP8_ENGINE_FLOORS): fp8 0.164, bf16 0.895 / 0.775; with the bias in the accumulator on top: fp8 0.141,
    # bf16 0.912 / 0.789 -- the strict mAP@.5:.95 of a random-weight network is the most sensitive of these rates to which way each value rounds
    assert r8["map50"] > 0.113 and r16["map50"] > 0.72 and r16["map"] > 0.62, (r8["map50"], r16["map50"], r16["map"])


@pytest.mark.parametrize("name", ["s_1280", "l_640", "l_1280"])
def test_fp8_against_sampled_reference_fixture(name):
    """The fp8 engine against the REFERENCE's own rows at the full sizes (tests/golden/detectors_sampled.npz: outputs of the reference
    classes, 4 096 sampled rows per case): class argmax and IoU of the decoded boxes on rows where the reference is confident,
    objectness error, calibrated on the case's own frames (16 where the batch has them).  Rates are printed next to the bf16 engine's
    (test_gpu_detector.py); the floors are 0.9 x the values measured in round 3 (profiles/r03_*_parity_margins.json)."""
    from cases import DETECTOR_CASES, variant_of
    G = os.path.join(ROOT, "tests", "golden")
    S = np.load(os.path.join(G, "detectors_sampled.npz"))
    case = [c for c in DETECTOR_CASES if c["name"] == name][0]
    h, w = case["hw"]
    frames = seeded_scene(case["batch"], h, w, case["seed"])
    x = torch.from_numpy(frames).cuda()
    out = {}
    for prec in ("bf16", "fp8"):
        m = _detector(variant_of(case), prec)
        if prec == "fp8":
            m.calibrate(x[:16])
        det, _ = m(x)
        assert bool(torch.isfinite(det).all())
        flat = det.cpu().numpy().reshape(-1, det.shape[-1])
        got, ref = flat[S[f"{name}.rows"]], S[f"{name}.det_rows"]
        conf = ref[:, 4] > 0.25
        cls = float((got[conf, 5:].argmax(-1) == ref[conf, 5:].argmax(-1)).mean())
        ri = row_iou(got, ref, conf)
        out[prec] = dict(cls=cls, row_iou=float(ri.mean()), row50=float((ri > 0.5).mean()), dobj=float(np.abs(got[:, 4] - ref[:, 4]).mean()), rows=int(conf.sum()))
        record_agreement(f"{name} {prec} vs reference rows (fp8 file)", cls_agree=cls, row_iou_mean=float(ri.mean()), row_iou_gt50=float((ri > 0.5).mean()),
                         mean_dobj=out[prec]["dobj"], confident_rows=int(conf.sum()))
    f = out["fp8"]
    lo = FP8_REF_FLOORS[name]
    assert f["rows"] > 20
    assert f["cls"] >= lo["cls"] and f["row_iou"] >= lo["row_iou"] and f["row50"] >= lo["row50"] and f["dobj"] <= lo["dobj"], (out, lo)


# 0.9 x the round-3 measurements (objectness error: 1.1 x)
FP8_REF_FLOORS = {   # measured (profiles/r03_parity_margins.json): cls / row IoU / rows with IoU > 0.5 / mean |d obj|
    "s_1280": dict(cls=0.758, row_iou=0.588, row50=0.689, dobj=0.046),      # 0.843 / 0.654 / 0.766 / 0.0418 (bf16: 0.986 / 0.954 / 0.995 / 0.0031)
    "l_640": dict(cls=0.669, row_iou=0.348, row50=0.321, dobj=0.068),       # 0.744 / 0.387 / 0.357 / 0.0617 (bf16: 0.972 / 0.864 / 0.950 / 0.0055)
    "l_1280": dict(cls=0.711, row_iou=0.367, row50=0.347, dobj=0.066),      # 0.790 / 0.408 / 0.386 / 0.0598 (bf16: 0.975 / 0.879 / 0.962 / 0.0050)
}


def test_fp8_config5_shard_b32_1536_deterministic_and_batch_independent():
    """BASELINE.json configs[4], one GPU's shard at its own size: skyeye_l fp8, B = 32 frames of 1536 x 1536 (145 152 rows each):
    two runs bit-identical, frames of the batch equal the same frames run alone (as a batch of 2), all values finite, probabilities in
    [0, 1]; the post-NMS boxes of both runs are identical too."""
    from skyeye.utils.metrics import nms_raw
    m = _detector("skyeye_l", "fp8")
    x = torch.from_numpy(seeded_scene(32, 1536, 1536, 9)).cuda()
    m.calibrate(x[:16])
    a, _ = m(x, return_raw=False)
    ra, ca = nms_raw(a, 0.25, 0.45)
    a = a.clone()
    b, _ = m(x, return_raw=False)
    assert a.shape == (32, 145152, 15)
    assert bool(torch.isfinite(a).all())
    assert torch.equal(a, b), f"two runs differ in {int((a != b).sum())} values"
    rb, cb = nms_raw(b, 0.25, 0.45)
    assert torch.equal(ca, cb) and torch.equal(ra, rb)
    pair, _ = m(x[30:], return_raw=False)
    assert torch.equal(pair[0], a[30]) and torch.equal(pair[1], a[31]), "frames 30 / 31 of the batch differ from the same frames run as a batch of 2"
    assert float(a[..., 4:].min()) >= 0.0 and float(a[..., 4:].max()) <= 1.0


def test_fp8_full_size_1536_deterministic_and_batch_independent():
    """Config 5's geometry: skyeye_l, 1536 x 1536 (145 152 rows per frame)."""
    m = _detector("skyeye_l", "fp8")
    x = torch.from_numpy(seeded_scene(4, 1536, 1536, 5)).cuda()
    m.calibrate(x[:2])
    a, _ = m(x, return_raw=False)
    b, _ = m(x, return_raw=False)
    assert a.shape == (4, 145152, 15)
    assert bool(torch.isfinite(a).all())
    assert torch.equal(a, b), f"two runs differ in {int((a != b).sum())} values"
    one, _ = m(x[3:], return_raw=False)
    assert torch.equal(one[0], a[3]), "frame 3 of the batch differs from the frame run alone"
    assert float(a[..., 4:].min()) >= 0.0 and float(a[..., 4:].max()) <= 1.0


def test_fp8_large_grid_determinism_two_workgroups_per_cu():
    """Every fp8 kernel family under the load that exposed the round-1 store hazard: B = 16, several tiles per workgroup,
    three runs bit-identical and equal to the emulation of the arithmetic."""
    for cin, cout, k, s, hw in [(128, 128, 3, 1, 80), (64, 64, 3, 1, 160), (128, 128, 1, 1, 80), (128, 256, 3, 2, 80)]:
        m = load_seeded(M.ConvolutionBlock(cin, cout, k, s), 41).set_precision("fp8")
        x = torch.from_numpy(seeded_input("fp8.det.%d" % cin, (16, cin, hw, hw), 4, -2.0, 2.0)).cuda()
        ys = [m(x) for _ in range(3)]
        assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2]), (cin, cout, k, s)
        ref, s_out = _emulate_conv(m._engine([x]), x.cpu().numpy(), k, s)
        y = ys[0].cpu().numpy()
        diff = np.abs(y - ref)
        step = np.maximum(np.abs(ref), s_out * 2.0 ** -6) * 0.126
        assert (diff <= step).all() and (diff == 0).mean() > 0.97, (cin, cout, k, s, float((diff == 0).mean()))
