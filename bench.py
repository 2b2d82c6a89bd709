#!/usr/bin/env python3
"""bench.py -- frames/s of the SkyEye detection hot path on MI355X.

One "step" = one pass of the whole path over one batch of synthetic frames already resident in HBM:
uint8 frames -> SkyEyeDetector.forward (backbone, neck, heads, decode) -> non_max_suppression, through the
drop-in Python API (skyeye.core.models / skyeye.utils.metrics), i.e. through the C ABI of libskyeye_hip.so.

    python bench.py                                  # 1 GPU, BASELINE.json configs[1]: skyeye_s bf16 B=32 @1280x1280
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, every rank runs the same per-GPU batch (weak scaling, images are independent units),
then an RCCL all-gather of the fixed-capacity box buffers so every rank holds all results (BASELINE north_star).
Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "skyeye-aerial-object-detection-using-yolo_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}     # dense MFMA peaks, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBPS = 8000.0                            # HBM3E spec peak (same table)
GFLOP_PER_FRAME = {("skyeye_s", 1280): 83.0, ("skyeye_l", 1280): 459.7, ("skyeye_s", 640): 20.75,
                   ("skyeye_l", 1536): 661.9}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="skyeye_s")
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step")
    ap.add_argument("--size", type=int, default=1280)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--conf", type=float, default=0.25)
    ap.add_argument("--iou", type=float, default=0.45)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def build_model(name, precision, device):
    """name: a tests/golden/cases.py variant -- skyeye_s / _m / _l, skyeye_s_enh (cross-layer attention), skyeye_s_ha
    (config 3: windowed attention on P3 / P4 + transformer layer on P5 ahead of the detection convs)."""
    from helpers import build_detector, detector_params, variant_cfg, variant_enhanced
    P = detector_params(name)
    model = build_detector(variant_cfg(name), variant_enhanced(name))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()}, strict=True)
    model.eval().set_precision(precision)
    return model, P


def calibrate_objectness(model, x, target=0.01, conf=0.25):
    """Shift the objectness biases so that ~1% of the boxes pass conf (realistic NMS load)."""
    _, raw = model(x[:2])
    obj = torch.cat([r[..., 4].reshape(-1) for r in raw]).float()
    k = max(1, int(obj.numel() * target))
    q = torch.topk(obj, k).values[-1].item()
    shift = float(np.log(conf / (1 - conf))) - q
    no = raw[0].shape[-1]
    with torch.no_grad():
        for layer in model.detection_head.detection_layers:
            layer.bias.view(-1, no)[:, 4] += shift
    model.refresh_weights()
    return shift


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the SkyEye engine has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)      # nccl == RCCL on ROCm

    from skyeye.utils.metrics import nms_raw
    from skyeye.distributed import all_gather_detections

    model, P = build_model(a.model, a.precision, dev)
    B, S = a.batch, a.size
    frames_np = np.random.default_rng(rank).integers(0, 256, size=(B, 3, S, S), dtype=np.uint8)
    x = torch.from_numpy(frames_np).to(dev)
    shift = calibrate_objectness(model, x, 0.01, a.conf)

    def step():
        det, _raw = model(x)                                                    # detector.py:300-324
        rows, counts = nms_raw(det, a.conf, a.iou, max_detections=300)          # metrics.py:361-457, no host sync
        if world > 1:
            rows, counts = all_gather_detections(rows, counts)                  # RCCL over xGMI
        return rows, counts

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        rows, counts = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kept_mean = float(counts.float().mean().item())

    # per-batch latency distribution (p50), each step individually synchronised
    lat = []
    for _ in range(min(a.steps, 20)):
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize(dev)
        lat.append((time.perf_counter() - t1) * 1e3)
    lat.sort()

    frames = world * B * a.steps
    fps = frames / dt
    out = {
        "metric": "frames/sec @1280x1280 + p50 latency, skyeye_s 1-GPU and skyeye_l 8-GPU",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
        "p50_latency_ms": round(lat[len(lat) // 2], 3), "p99_latency_ms": round(lat[min(len(lat) - 1, int(0.99 * len(lat)))], 3),
        "config": {"workload": f"{a.model} {a.precision} batch={B}/GPU @{S}x{S}: uint8 frames in HBM -> backbone+neck+heads+"
                               f"decode -> NMS(conf {a.conf}, iou {a.iou}, max_det 300)"
                               + (" -> RCCL all-gather of boxes" if world > 1 else ""),
                   "global_batch": world * B, "frames_per_gpu": B, "image_size": S, "candidates_target": "1% > conf",
                   "mean_boxes_kept_per_image": round(kept_mean, 1), "parallelism": f"dp{world} (independent images)"},
    }

    if rank == 0 and not a.no_roofline:
        from skyeye import _native as N
        h = model._engine([x])
        outs = [torch.empty(s, dtype=torch.float32, device=dev) for s in h.output_shapes()]
        stream = torch.cuda.current_stream(dev).cuda_stream
        prof = h.profile_forward([N.buffer_from_tensor(x)], [N.buffer_from_tensor(t) for t in outs], stream, iters=3)
        by = {}
        for i, (ms, fl, tag) in enumerate(prof):
            e = by.setdefault(tag, [0.0, 0.0, 0, 0.0])
            e[0] += ms; e[1] += fl; e[2] += 1; e[3] += h.op_bytes(i)
        conv = {t: v for t, v in by.items() if t // 10000 == 2}      # OP_CONV
        dom_tag, dom = max(conv.items(), key=lambda kv: kv[1][0])
        variant = dom_tag % 10000
        tname = "bf16" if a.precision == "bf16" else "float"
        kname = (f"conv_halo_kernel<{tname}> stride-2 halo tile + weight ring, N_blk {variant % 1000}" if variant >= 6000 else
                 f"conv_halo_small_kernel<{tname}> narrow-input halo tile, N_blk {variant % 1000}" if variant >= 5000 else
                 f"conv_halo_kernel<{tname}> halo tile + weight ring, N_blk {variant % 1000}" if variant >= 4000 else
                 f"conv_stream_kernel<{tname}> weight-ring, N_blk {variant % 1000}" if variant >= 3000 else
                 f"conv_stream_kernel<{tname}> resident weights, N_blk {variant % 1000}" if variant >= 2000 else
                 f"conv_igemm_kernel<{tname}> N tile {variant % 1000}")
        conv_ms = sum(v[0] for v in conv.values()); conv_fl = sum(v[1] for v in conv.values())
        tot_ms = sum(v[0] for v in by.values())
        peak = PEAK_TFLOPS[a.precision]
        ach = dom[1] / (dom[0] * 1e-3) / 1e12
        traffic, traffic_src = None, None
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json"))):     # latest PMC pass (tools/pmc_traffic.sh)
            try:
                tv = json.load(open(f))["by_variant"].get(str(variant))
                if tv and a.model == "skyeye_s" and B == 32 and S == 1280 and a.precision == "bf16":
                    traffic, traffic_src = round(tv["hbm_bytes_per_launch"]), os.path.relpath(f, ROOT)
            except Exception:  # noqa: BLE001
                pass
        hbm_gbps = dom[3] / (dom[0] * 1e-3) / 1e9                         # algorithmic in+out(+residual) bytes per launch / time
        frac_mfma, frac_hbm = ach / peak, hbm_gbps / PEAK_HBM_GBPS
        hbm_bound = frac_hbm > frac_mfma                                   # report the roof this kernel group is closer to
        out["roofline"] = {
            "bound": "hbm" if hbm_bound else "mfma", "kernel": kname,
            "achieved": round(hbm_gbps if hbm_bound else ach, 2), "peak": PEAK_HBM_GBPS if hbm_bound else peak,
            "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": round(max(frac_hbm, frac_mfma), 4), "traffic": traffic,
            "traffic_source": traffic_src, "algorithmic_bytes_per_launch": round(dom[3] / dom[2]),
            "hbm_gbps_algorithmic": round(hbm_gbps, 1), "frac_hbm": round(frac_hbm, 4),
            "tflops": round(ach, 2), "frac_mfma": round(frac_mfma, 4),
            "launches_per_step": dom[2], "avg_launch_ms": round(dom[0] / dom[2], 4),
            "flops_per_launch": dom[1] / dom[2],
            "all_conv_tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2), "conv_ms_per_step": round(conv_ms, 3),
            "graph_ms_per_step": round(tot_ms, 3), "graph_launches": len(prof),
            "conv_variants": {str(t % 10000): {"launches": v[2], "ms": round(v[0], 3), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1)}
                              for t, v in sorted(conv.items())},
            "graph_gflop_per_frame": round(sum(v[1] for v in by.values()) / B / 1e9, 2),
            "baseline_gflop_per_frame": GFLOP_PER_FRAME.get((a.model, S)),
            "end_to_end_tflops": round(fps / world * sum(v[1] for v in by.values()) / B / 1e12, 2),
        }

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # CPU baseline: the oracle (plain C + OpenMP port of the reference path) on the host cores of this box,
        # same weights, same kind of frames, bounded sample.  Checker code only -- never on the product path.
        from oracle import skyeye_oracle as O
        O.set_threads(min(64, os.cpu_count() or 1))          # beyond ~64 threads the band-parallel C port stops scaling
        Pc = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in model.state_dict().items()}
        from helpers import variant_cfg, variant_enhanced
        vcfg = variant_cfg(a.model)
        nc = vcfg["nc"]
        okw = dict(enhanced=variant_enhanced(a.model), head_attention=bool(vcfg.get("head_attention", False)))
        xs = frames_np[:1].astype(np.float32) / np.float32(255.0)
        t1 = time.perf_counter()
        d, _ = O.detector_forward(Pc, xs, nc, **okw)
        O.non_max_suppression(d, a.conf, a.iou)
        one = time.perf_counter() - t1
        n = int(max(1, min(8, 20.0 / max(one, 1e-3))))
        t1 = time.perf_counter()
        for i in range(n):
            d, _ = O.detector_forward(Pc, frames_np[i % B:i % B + 1].astype(np.float32) / np.float32(255.0), nc, **okw)
            O.non_max_suppression(d, a.conf, a.iou)
        cpu_dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(n / cpu_dt, 4), "unit": "frames/s", "cores": O.threads(), "kind": "port",
                               "sample": f"{n} frame(s) of the same workload ({a.model} fp32 @{S}x{S}, forward+decode+NMS), "
                                         f"oracle/ C+OpenMP port, {os.cpu_count()} host CPUs visible"}

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
