#!/usr/bin/env python3
"""Generate the golden fixtures by running the REFERENCE's own PyTorch classes.

Runs only in the build container (needs /root/reference); the GPU box and the
test-suite never import the reference -- they read the ``*.npz`` this writes.

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py blocks nms # a subset

What is executed from the reference, unmodified (file:line under /root/reference):
  * skyeye/core/models/blocks.py     ConvolutionBlock :10-41, BottleneckBlock :69-90,
                                     CSPBlock :93-123, SPPBlock :126-149, FocusBlock :152-182
  * skyeye/core/models/attention.py  ChannelAttention :11-60, SpatialAttention :63-98,
                                     CombinedAttention :101-130, CrossLayerAttention :133-241,
                                     TransformerLayer :244-309, WindowedSelfAttention :312-399
  * skyeye/core/models/backbone.py   Backbone :12-99, SkyEyeBackbone :119-159
  * skyeye/core/models/detector.py   DetectionHead :18-145, FeatureNeck :148-231
  * skyeye/utils/metrics.py          non_max_suppression :361-457 (wrapper only)
  * skyeye/utils/torch_utils.py      scale_img :262-288

How the classes are composed (SURVEY.md Appendix A -- the reference's own
``SkyEyeDetector`` cannot be constructed or run):
  D1  FeatureNeck(true_channels, width_multiple=1.0)
  D2  neck/head channel counts are read from the backbone's real outputs
  D3  ``SkyEyeDetector._initialize_weights`` is never called (weights are seeded)
  D4  EnhancedSkyEyeDetector's CrossLayerAttention gets key/value projections
      ``key_channels -> query_channels`` (instance surgery, file untouched)
  D6  ``torchvision.ops.nms`` (absent, un-vendored, unpinned: requirements.txt:2
      says only ``torchvision>=0.8.1``) is replaced by the restated greedy NMS
      below, injected as ``metrics.torchvision``.  -> NMS core: PARITY UNPINNED;
      the wrapper around it is pinned.
"""
import hashlib
import json
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = os.environ.get("SKYEYE_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from cases import BLOCK_CASES, DETECTOR_CASES, MODELS, N_SAMPLED_ROWS, NMS_CASES, WSEED, variant_of  # noqa: E402
from nms_inputs import make_predictions  # noqa: E402
from seeded import seeded_input, seeded_scene, seeded_tensor  # noqa: E402

warnings.filterwarnings("ignore", message="torch.meshgrid")
torch.set_grad_enabled(False)
torch.set_num_threads(8)

from skyeye.core.models import attention as ref_attention  # noqa: E402
from skyeye.core.models import backbone as ref_backbone  # noqa: E402
from skyeye.core.models import blocks as ref_blocks  # noqa: E402
from skyeye.core.models import detector as ref_detector  # noqa: E402


# --------------------------------------------------------------------------- helpers
def load_seeded(module, seed, prefix=""):
    """Push seeded.py tensors into a reference module through load_state_dict."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if k.rsplit(".", 1)[-1] == "relative_position_index":   # derived index buffer, attention.py:342-353
            new[k] = v
            continue
        new[k] = torch.from_numpy(seeded_tensor(prefix + k, tuple(v.shape), seed)).to(v.dtype).reshape(v.shape)
    module.load_state_dict(new, strict=True)
    return module.eval()


def make_cla_d4(query_channels, key_channels, region_size=2, heads=4):
    """D4: CLA whose key/value projections map key_channels -> query_channels."""
    m = ref_attention.CrossLayerAttention(query_channels, query_channels, region_size=region_size, heads=heads)
    m.key_projection = nn.Conv2d(key_channels, query_channels, kernel_size=1)
    m.value_projection = nn.Conv2d(key_channels, query_channels, kernel_size=1)
    return m


class ComposedDetector(nn.Module):
    """Backbone -> FeatureNeck -> DetectionHead -> process_detections with D1/D2/D3.

    Attribute names equal SkyEyeDetector's (detector.py:268-285) so the
    state-dict keys are the reference's.
    """

    def __init__(self, cfg, enhanced=False, head_attention=False):
        super().__init__()
        self.backbone = ref_backbone.SkyEyeBackbone(cfg["base_channels"], cfg["depth_multiple"], cfg["width_multiple"])
        feats, _wrong_channels = self.backbone(torch.zeros(1, 3, 64, 64))
        true_channels = [f.shape[1] for f in feats]                       # D2
        self.neck = ref_detector.FeatureNeck(true_channels, width_multiple=1.0)  # D1
        self.detection_head = ref_detector.DetectionHead(cfg["nc"], cfg.get("anchors"), self.neck.out_channels)
        self.enhanced = enhanced
        if enhanced:  # detector.py:457-469 with D4
            c3, c4, c5 = self.neck.out_channels
            self.cross_attention_p5_p4 = make_cla_d4(c4, c5)
            self.cross_attention_p4_p3 = make_cla_d4(c3, c4)
        self.has_head_attention = head_attention
        if head_attention:   # D5 (build-defined call site): attention.py:244-399 modules ahead of the detection convs
            c3, c4, c5 = self.neck.out_channels
            self.head_attention = nn.ModuleDict({"p3": ref_attention.WindowedSelfAttention(c3, 8, c3 // 32),
                                                 "p4": ref_attention.WindowedSelfAttention(c4, 8, c4 // 32),
                                                 "p5": ref_attention.TransformerLayer(c5, 8)})

    @staticmethod
    def windowed(m, x, ws=8):
        """window_partition -> WindowedSelfAttention -> window_reverse on a [B, C, H, W] map"""
        B, C, H, W = x.shape
        t = x.permute(0, 2, 3, 1).reshape(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
        t = m(t)
        t = t.reshape(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
        return t.permute(0, 3, 1, 2).contiguous()

    def forward(self, x):
        feats, _ = self.backbone(x)                                       # detector.py:311
        if self.enhanced:                                                 # detector.py:485-491
            p3, p4, p5 = self.neck(feats)
            p4e = self.cross_attention_p5_p4(p4, p5) + p4
            p3e = self.cross_attention_p4_p3(p3, p4e) + p3
            neck = [p3e, p4e, p5]
        else:
            neck = self.neck(feats)                                       # detector.py:314
        if self.has_head_attention:
            neck = [self.windowed(self.head_attention["p3"], neck[0]), self.windowed(self.head_attention["p4"], neck[1]),
                    self.head_attention["p5"](neck[2])]
        outputs = self.detection_head(neck)                               # detector.py:317
        raw = [o.clone() for o in outputs]
        det = self.detection_head.process_detections(outputs, x.shape[2:])  # detector.py:321
        return det, raw


def build_block(case):
    kind, args = case["kind"], dict(case["args"])
    if kind == "CrossLayerAttentionD4":
        return make_cla_d4(**args)
    for mod in (ref_blocks, ref_attention, ref_backbone, ref_detector):
        if hasattr(mod, kind):
            return getattr(mod, kind)(**args)
    raise KeyError(kind)


def run_block(case):
    m = load_seeded(build_block(case), case["seed"])
    ins = {k: torch.from_numpy(seeded_input(case["name"] + "." + k, shp, case["seed"], lo, hi))
           for k, (shp, lo, hi) in case["inputs"].items()}
    kind = case["kind"]
    if kind in ("FeatureNeck",):
        outs = m([ins["p3"], ins["p4"], ins["p5"]])
    elif kind == "DetectionHead":
        feats = [ins[k] for k in sorted(ins)]
        raw = m(feats)
        rawc = [r.clone() for r in raw]
        det = m.process_detections(raw, case["input_shape"])
        outs = [det] + rawc
    elif kind in ("CrossLayerAttention", "CrossLayerAttentionD4"):
        outs = [m(ins["q"], ins["k"])]
    elif kind == "WindowedSelfAttention":
        outs = [m(ins["x"], ins.get("mask"))]
    else:
        outs = m(ins["x"])
        if torch.is_tensor(outs):
            outs = [outs]
    return [o.numpy().astype(np.float32) for o in outs]


# --------------------------------------------------------------------------- NMS
def greedy_nms(boxes, scores, iou_threshold):
    """Restatement of torchvision.ops.nms' documented behaviour:
    visit boxes by descending score (ties: lower index first -- the build's
    choice, torchvision leaves it unspecified), keep a box unless an already
    kept one has IoU > threshold with it; IoU = inter / (a_i + a_j - inter),
    areas (x2-x1)*(y2-y1), fp32 arithmetic, strict '>'.  Returns kept indices
    in visiting order."""
    b = boxes.detach().cpu().numpy().astype(np.float32)
    s = scores.detach().cpu().numpy().astype(np.float32)
    n = b.shape[0]
    order = np.lexsort((np.arange(n), -s.astype(np.float64)))
    x1, y1, x2, y2 = (b[order, i] for i in range(4))
    area = (x2 - x1) * (y2 - y1)
    dead = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_threshold)
    with np.errstate(all="ignore"):
        for i in range(n):
            if dead[i]:
                continue
            keep.append(order[i])
            if i + 1 == n:
                break
            w = np.maximum(np.float32(0), np.minimum(x2[i], x2[i + 1:]) - np.maximum(x1[i], x1[i + 1:]))
            h = np.maximum(np.float32(0), np.minimum(y2[i], y2[i + 1:]) - np.maximum(y1[i], y1[i + 1:]))
            inter = w * h
            iou = inter / (area[i] + area[i + 1:] - inter)
            dead[i + 1:] |= iou > thr
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


def load_reference_metrics():
    """skyeye/utils/__init__.py cannot be imported (cv2, undefined names); load
    metrics.py under a stub parent package instead."""
    import importlib
    pkg = types.ModuleType("skyeye.utils")
    pkg.__path__ = [os.path.join(REF, "skyeye", "utils")]
    sys.modules["skyeye.utils"] = pkg
    metrics = importlib.import_module("skyeye.utils.metrics")
    tv = types.SimpleNamespace(ops=types.SimpleNamespace(nms=greedy_nms))
    metrics.torchvision = tv                                             # D6
    return metrics


# --------------------------------------------------------------------------- writers
def gen_blocks():
    out = {}
    for case in BLOCK_CASES:
        outs = run_block(case)
        for i, o in enumerate(outs):
            out[f"{case['name']}.out{i}"] = o
        print(f"  block {case['name']:28s} -> {[o.shape for o in outs]}")
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **out)


def calibrated_detector(variant, calib, size=192, scenes=24):
    """Seeded weights + BatchNorm running statistics measured on one calibration batch of 24 structured 192x192
    scenes (momentum 1.0: running stats := batch stats), so that layer outputs are O(1) like in a trained network."""
    ha = variant.endswith("_ha")
    base = variant[:-3] if ha else variant
    enhanced = base.endswith("_enh")
    cfg = MODELS[base[:-4] if enhanced else base]
    m = load_seeded(ComposedDetector(cfg, enhanced=enhanced, head_attention=ha), WSEED[variant])
    bns = [mod for mod in m.modules() if isinstance(mod, nn.BatchNorm2d)]
    for bn in bns:
        bn.momentum = 1.0
    m.train()
    m(torch.from_numpy(seeded_scene(scenes, size, size, 777)).float() / 255.0)
    m.eval()
    for k, v in m.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            calib[f"{variant}:{k}"] = v.numpy().astype(np.float32)
    return m


def gen_detectors():
    full, sampled, calib, models = {}, {}, {}, {}
    for case in DETECTOR_CASES:
        cfg = MODELS[case["model"]]
        variant = variant_of(case)
        if variant not in models:
            models[variant] = calibrated_detector(variant, calib)
        m = models[variant]
        h, w = case["hw"]
        frames = seeded_scene(case["batch"], h, w, case["seed"])
        x = torch.from_numpy(frames).float() / 255.0                      # validate.py:236-238
        det, raw = m(x)
        det = det.numpy().astype(np.float32)
        raw = [r.numpy().astype(np.float32) for r in raw]
        name = case["name"]
        if case["store"] == "full":
            full[f"{name}.det"] = det
            for i, r in enumerate(raw):
                full[f"{name}.raw{i}"] = r
        else:
            flat = det.reshape(-1, det.shape[-1])
            rows = np.random.default_rng(case["seed"]).choice(flat.shape[0], N_SAMPLED_ROWS, replace=False)
            rows.sort()
            sampled[f"{name}.rows"] = rows.astype(np.int64)
            sampled[f"{name}.det_rows"] = flat[rows]
            sampled[f"{name}.sha256"] = np.frombuffer(hashlib.sha256(det.tobytes()).digest(), dtype=np.uint8)
            sampled[f"{name}.absmean"] = np.asarray(np.abs(flat.astype(np.float64)).mean(axis=0), dtype=np.float64)
            for i, r in enumerate(raw):
                rf = r.reshape(-1, r.shape[-1])
                rr = np.random.default_rng(case["seed"] + 1 + i).choice(rf.shape[0], min(512, rf.shape[0]), replace=False)
                rr.sort()
                sampled[f"{name}.raw{i}_rows"] = rr.astype(np.int64)
                sampled[f"{name}.raw{i}_vals"] = rf[rr]
        obj = det[..., 4]
        print(f"  detector {name:14s} det {det.shape}  obj>0.25: {(obj > 0.25).mean():.4f}  "
              f"|xy|max {np.abs(det[..., :2]).max():.1f} wh max {det[..., 2:4].max():.1f}")
    np.savez_compressed(os.path.join(HERE, "bn_calib.npz"), **calib)
    np.savez_compressed(os.path.join(HERE, "detectors_full.npz"), **full)
    np.savez_compressed(os.path.join(HERE, "detectors_sampled.npz"), **sampled)


def gen_head_attention():
    """D5 wiring: ComposedDetector(head_attention=True) = reference WindowedSelfAttention (attention.py:312-399) on the
    8x8 windows of P3 / P4 and reference TransformerLayer (:244-309) on P5, then DetectionHead."""
    from cases import HA_CASES
    full, calib = {}, {}
    m = calibrated_detector("skyeye_s_ha", calib, size=256, scenes=8)
    for case in HA_CASES:
        h, w = case["hw"]
        x = torch.from_numpy(seeded_scene(case["batch"], h, w, case["seed"])).float() / 255.0
        det, raw = m(x)
        if case["store"] == "sampled":           # full size: SHA-256 of the tensor + sampled rows, like detectors_sampled.npz
            name = case["name"]
            detn = det.numpy().astype(np.float32)
            flat = detn.reshape(-1, detn.shape[-1])
            rows = np.random.default_rng(case["seed"]).choice(flat.shape[0], N_SAMPLED_ROWS, replace=False)
            rows.sort()
            full[f"{name}.rows"] = rows.astype(np.int64)
            full[f"{name}.det_rows"] = flat[rows]
            full[f"{name}.sha256"] = np.frombuffer(hashlib.sha256(detn.tobytes()).digest(), dtype=np.uint8)
            full[f"{name}.absmean"] = np.asarray(np.abs(flat.astype(np.float64)).mean(axis=0), dtype=np.float64)
            for i, r in enumerate(raw):
                rf = r.numpy().astype(np.float32).reshape(-1, r.shape[-1])
                rr = np.random.default_rng(case["seed"] + 1 + i).choice(rf.shape[0], min(512, rf.shape[0]), replace=False)
                rr.sort()
                full[f"{name}.raw{i}_rows"] = rr.astype(np.int64)
                full[f"{name}.raw{i}_vals"] = rf[rr]
            print(f"  head-attention {name} det {tuple(det.shape)} (sampled)")
            continue
        full[f"{case['name']}.det"] = det.numpy().astype(np.float32)
        for i, r in enumerate(raw):
            full[f"{case['name']}.raw{i}"] = r.numpy().astype(np.float32)
        print(f"  head-attention {case['name']} det {tuple(det.shape)} obj>0.25 {(det[..., 4] > 0.25).float().mean():.4f} raw absmax {max(float(r.abs().max()) for r in raw):.2f}")
    np.savez_compressed(os.path.join(HERE, "bn_calib_ha.npz"), **calib)
    np.savez_compressed(os.path.join(HERE, "detectors_ha.npz"), **full)


def gen_nms():
    metrics = load_reference_metrics()
    out = {}
    for case in NMS_CASES:
        pred = make_predictions(case["nc"], case["batch"], case["n"], case["seed"], ties=case.get("ties", False),
                                distinct_scores=(case["name"] == "over_cap"))
        res = metrics.non_max_suppression(torch.from_numpy(pred), **case["kwargs"])   # metrics.py:361
        counts = np.asarray([r.shape[0] for r in res], dtype=np.int64)
        cols = max([r.shape[1] for r in res if r.shape[0]] + [0])
        rows = [r.numpy().astype(np.float32).reshape(-1, cols) for r in res if r.shape[0]]
        out[f"{case['name']}.counts"] = counts
        out[f"{case['name']}.rows"] = np.concatenate(rows, 0) if rows else np.zeros((0, cols or 6), np.float32)
        print(f"  nms {case['name']:22s} counts {counts.tolist()} cols {cols}")
    np.savez_compressed(os.path.join(HERE, "nms.npz"), **out)


def eval_inputs():
    """Seeded inputs of the evaluation-accounting fixtures."""
    r = np.random.default_rng(20240611)

    def boxes(n, degenerate=0):
        xy = r.uniform(0, 600, (n, 2)).astype(np.float32)
        wh = r.uniform(2, 200, (n, 2)).astype(np.float32)
        b = np.concatenate([xy, xy + wh], 1)
        b[:degenerate, 2:] = b[:degenerate, :2]            # zero-area boxes
        return b

    a, b = boxes(37, 2), boxes(53, 1)
    b[5] = a[7]                                            # an exact duplicate: IoU ~ 1
    curves = []
    for n in (1, 17, 400):
        rec = np.sort(r.uniform(0, 1, n)).astype(np.float64)
        prec = r.uniform(0, 1, n).astype(np.float64)
        curves.append((rec, prec))
    n, m, nc = 600, 240, 10
    conf = r.uniform(0, 1, n)
    tp = r.uniform(0, 1, (n, 10)) < (0.2 + 0.6 * conf[:, None])        # better-scored detections are right more often
    tp = np.logical_and.accumulate(tp[:, ::-1], axis=1)[:, ::-1] | tp   # keep it a plausible monotone-ish matrix
    pred_cls = r.integers(0, nc - 1, n).astype(np.float64)             # class nc-1 is never predicted
    target_cls = r.integers(1, nc, m).astype(np.float64)               # class 0 has no labels
    return a, b, curves, (tp, conf, pred_cls, target_cls)


def gen_eval():
    """box_iou metrics.py:17-44 (LITERAL: box1 is indexed as [4, N]), compute_ap :124-148, ap_per_class :151-225."""
    metrics = load_reference_metrics()
    a, b, curves, (tp, conf, pred_cls, target_cls) = eval_inputs()
    out = {"iou.a": a, "iou.b": b}
    out["iou.out"] = metrics.box_iou(torch.from_numpy(a.T.copy()), torch.from_numpy(b)).numpy()        # [37, 53]
    for k, (rec, prec) in enumerate(curves):
        ap, mpre, mrec = metrics.compute_ap(rec, prec)
        out[f"ap{k}.recall"], out[f"ap{k}.precision"] = rec, prec
        out[f"ap{k}.ap"], out[f"ap{k}.mpre"], out[f"ap{k}.mrec"] = np.float64(ap), mpre, mrec
    p, r_, ap, f1, cls = metrics.ap_per_class(tp, conf, pred_cls, target_cls)
    out.update({"apc.tp": tp, "apc.conf": conf, "apc.pred_cls": pred_cls, "apc.target_cls": target_cls,
                "apc.p": p, "apc.r": r_, "apc.ap": ap, "apc.f1": f1, "apc.classes": cls})
    np.savez_compressed(os.path.join(HERE, "eval.npz"), **out)
    print(f"  eval: iou {out['iou.out'].shape} max {out['iou.out'].max():.6f}; ap_per_class ap {ap.shape} mean {ap.mean():.4f}")


def gen_tta():
    """scale_img torch_utils.py:262-288, unmodified, on ``x.flip(f)`` (the caller's flip of YOLOv5's _forward_augment, which the
    reference's ``augment=`` flag, validate.py:245 / detect.py:140, stands for; the reference has no body for it)."""
    import importlib
    from cases import TTA_CASES
    pkg = types.ModuleType("skyeye.utils")
    pkg.__path__ = [os.path.join(REF, "skyeye", "utils")]
    sys.modules["skyeye.utils"] = pkg
    tu = importlib.import_module("skyeye.utils.torch_utils")
    out = {}
    for name, (B, H, W, ratio, flip, same, gs) in TTA_CASES.items():
        x = torch.from_numpy(seeded_input("tta." + name, (B, 3, H, W), 7))
        y = tu.scale_img(x.flip(flip) if flip else x, ratio, same_shape=same, gs=gs).numpy()
        if name.endswith("_big"):                     # keep the file small: shape + sampled rows
            out[name + ".shape"] = np.array(y.shape)
            out[name + ".rows"] = y[:, :, ::37].copy()
        else:
            out[name] = y
        print(f"  tta {name}: {tuple(x.shape)} -> {y.shape}")
    np.savez_compressed(os.path.join(HERE, "tta.npz"), **out)


def main(argv):
    what = set(argv) or {"blocks", "detectors", "nms", "eval", "ha", "tta"}
    if "tta" in what:
        gen_tta()
    if "eval" in what:
        gen_eval()
    if "ha" in what:
        gen_head_attention()
    if "blocks" in what:
        gen_blocks()
    if "detectors" in what:
        gen_detectors()
    if "nms" in what:
        gen_nms()
    meta = dict(torch=torch.__version__, numpy=np.__version__, reference=REF, threads=torch.get_num_threads())
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
