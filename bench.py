#!/usr/bin/env python3
"""bench.py -- frames/s of the SkyEye detection hot path on MI355X.

One "step" = one pass of the whole path over one batch of synthetic frames already resident in HBM:
uint8 frames -> SkyEyeDetector.forward (backbone, neck, heads, decode) -> non_max_suppression, through the
drop-in Python API (skyeye.core.models / skyeye.utils.metrics), i.e. through the C ABI of libskyeye_hip.so.

    python bench.py                                  # 1 GPU, BASELINE.json configs[1]: skyeye_s bf16 B=32 @1280x1280
    python bench.py --gpus N                         # spawns `python -m torch.distributed.run --nproc-per-node N` itself
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --model skyeye_s_ha              # config 3 (attention heads on)
    python bench.py --model skyeye_l                 # config 4, one GPU's shard
    python bench.py --model skyeye_l --precision fp8 --size 1536   # config 5, one GPU's shard

N > 1: one process per GPU, every rank runs the same per-GPU batch (weak scaling, images are independent units),
then an RCCL all-gather of the fixed-capacity box buffers so every rank holds all results (BASELINE north_star).
Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and `cpu_baseline`.

`roofline` is a fixed instrument: `frac` = END-TO-END MFMA fraction by SURVEY 8(d)'s formula,
frames/s per GPU x GFLOP/frame / dense MFMA peak of the dtype; `families` lists every kernel family of the forward with BOTH
its HBM fraction (algorithmic bytes / time / 8 TB/s) and its MFMA fraction, measured with hipEvents recorded inside
libskyeye_hip.so on the stream the kernels run on (sky_profile_forward); `dominant` is the family with the largest summed
time; `traffic` / per-family `pmc_bytes` come from the latest committed PMC pass (profiles/*pmc_traffic.json).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "skyeye-aerial-object-detection-using-yolo_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "fp8": 5000.0}   # dense MFMA peaks, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBPS = 8000.0                                          # HBM3E spec peak (same table)
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
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp8"])
    ap.add_argument("--conf", type=float, default=0.25)
    ap.add_argument("--iou", type=float, default=0.45)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--with-raw", action="store_true", help="also write the three raw detection levels (the reference's second return value)")
    ap.add_argument("--streams", type=int, default=2, help="run the batch as this many equal slices on parallel branches of the captured graph (each "
                    "slice has its own plan); 1 = the whole batch through one plan")
    ap.add_argument("--slices", default="", help="explicit slice sizes for the parallel streams, e.g. 20,12 (default: --streams equal slices)")
    ap.add_argument("--no-pipeline", action="store_true", help="NMS of a batch strictly behind its own forward pass (default: the NMS of batch k runs beside the "
                    "forward pass of batch k + 1, SkyEyeDetector.detect_nms_pipelined; every batch's NMS is inside the timed region)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short legs of configs 3, 4 (shard), 5 (shard) behind the headline measurement")
    return ap.parse_args()


def spawn_workers(a):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD (never exec: this process may
    already have touched the GPU) and relay its JSON line and exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line:
        print(line)
    elif r.stdout:
        sys.stderr.write(r.stdout[-4000:])
    raise SystemExit(r.returncode if r.returncode or line else 1)


def build_model(name, precision, device):
    """name: a tests/golden/cases.py variant -- skyeye_s / _m / _l, skyeye_s_enh (cross-layer attention), skyeye_s_ha
    (config 3: windowed attention on P3 / P4 + transformer layer on P5 ahead of the detection convs)."""
    import numpy as np
    import torch
    from helpers import build_detector, detector_params, variant_cfg, variant_enhanced
    P = detector_params(name)
    model = build_detector(variant_cfg(name), variant_enhanced(name))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()}, strict=True)
    model.eval().set_precision(precision)
    return model, P


def calibrate_objectness(model, x, target=0.01, conf=0.25):
    """Shift the objectness biases so that ~1% of the boxes pass conf (realistic NMS load)."""
    import numpy as np
    import torch
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


FAMILIES = [  # (name, kernel, predicate on the convolution variant tag recorded by launch_conv)
    ("deep3x3", "conv3x3_deep_kernel: 3x3 stride 1, Cin >= 256: two halo images, 4-stage weight ring, counted waits", lambda v: 4600 <= v < 4800),
    ("halo3x3", "conv_halo_kernel: 3x3 halo tile + weight ring (stride 1 and 2)", lambda v: 4000 <= v < 5000 or 6000 <= v < 7000),
    ("halo_narrow", "conv_halo_small_kernel: stem / narrow-input 3x3, K resident", lambda v: 5000 <= v < 6000),
    ("bneck128", "bneck128w_kernel (bf16) / bneck128w8_kernel (fp8): BottleneckBlock(128, 128) 1x1 -> 3x3 (+residual) in one kernel, two workgroups per CU", lambda v: v in (7128, 7256, 7257)),
    ("halo_cv1", "bneck64w_kernel (three workgroups per CU) / conv_halo_kernel<CV1>: BottleneckBlock(64, 64) 1x1 -> 3x3 (+residual) fused", lambda v: 7000 <= v < 8000),
    ("stem_down", "stem_down_kernel: frames -> FocusBlock 3x3 -> 3x3 stride 2 in one kernel", lambda v: 8000 <= v < 8500),
    ("csp_stage", "csp_stage_kernel: CSPBlock(64, 64, 1) = cv1|cv2 -> 1x1 -> 3x3 + shortcut -> cv3 in one kernel", lambda v: 8500 <= v < 9000),
    ("stream_resident", "conv_stream_kernel: 1x1 (and narrow 3x3), weights resident in LDS", lambda v: 2000 <= v < 3000),
    ("gemm1x1", "gemm1x1_kernel: large-K 1x1 as a 256 x 256-tile GEMM, both operands by LDS-DMA", lambda v: v == 3256),
    ("stream_ring", "conv_stream_kernel: large-K 1x1, weight ring", lambda v: 3000 <= v < 4000),
    ("cv3_head", "cv3_head_kernel: CSP cv3 (1x1 128->128) + detection level 0 (conv + decode) in one kernel", lambda v: v == 1628),
    ("tile", "conv_igemm_kernel / head_stream_kernel: detection levels (N = 45) + fallback shapes", lambda v: 1000 <= v < 2000),
]


def family_of(tag):
    if tag // 10000 != 2:
        return "non_conv"
    v = tag % 10000
    for name, _, pred in FAMILIES:
        if pred(v):
            return name
    return "conv_other"


def _native_build():
    """Hash of the sources libskyeye_hip.so was built from (sky_build_info) and whether it equals the working tree's."""
    from skyeye import _native
    src = _native.source_hash()
    return {"library_sources": _native.build_info(), "matches_tree": None if src is None else src == _native.build_info()}


OTHER_CONFIGS = [   # BASELINE.json configs[2], [3] (one GPU's shard), [4] (one GPU's shard): short legs behind the headline measurement;
    # last: the headline workload on the fp32 (exact) engine -- the engine the 1e-4 parity tests pin -- so that it has a driver-visible number
    ("skyeye_s_ha", "bf16", 1280), ("skyeye_l", "bf16", 1280), ("skyeye_l", "fp8", 1536), ("skyeye_s", "fp32", 1280)]


def plan_gflop_per_frame(model, x):
    """Algorithmic GFLOP per frame of the planned graph (sky_plan_stats: 2 * MAC over every convolution / linear layer AND the
    attention products of the head-attention variant), from the plan the leg really ran (a slice plan when the batch is sliced)."""
    nsl = model.__dict__.get("_slices", 1)
    xi = model._prepare_input(x)
    if nsl > 1 and model._sliceable([xi], nsl):
        sizes = model.__dict__.get("_slice_sizes") or (xi.shape[0] // nsl,) * nsl
        part = xi[:sizes[0]]
        return model._engine_entry([part], None, slot=1)[1].stats()["flops"] / part.shape[0] / 1e9
    return model._engine([xi]).stats()["flops"] / xi.shape[0] / 1e9


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_workers(a)
    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} (or run "
                         f"`python bench.py --gpus {a.gpus}` and let it spawn the workers)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the SkyEye engine has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    ranks_seen = 1
    # readiness switch (one GPU): run the N > 1 code path -- exchange blocks written by the NMS kernel, the RCCL all-gather of every step, the
    # per-rank reductions -- on a world-size-1 nccl group, so that the first multi-GPU run is not the first time this code executes
    force_ex = world == 1 and bool(os.environ.get("SKY_BENCH_FORCE_EXCHANGE"))
    if world > 1 or force_ex:
        import torch.distributed as dist
        # RCCL prints a version banner on STDOUT when the first communicator is made: the contract is ONE JSON line there, so the
        # C-level stdout goes to stderr until the first collective has run
        sys.stdout.flush()
        saved_out = os.dup(1)
        os.dup2(2, 1)
        if force_ex:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)      # nccl == RCCL on ROCm
        try:
            one = torch.ones(1, device=dev)
            dist.all_reduce(one)
            ranks_seen = int(one.item())
        finally:
            sys.stdout.flush()
            os.dup2(saved_out, 1)
            os.close(saved_out)
    multi = world > 1 or force_ex

    from skyeye.utils.metrics import nms_raw
    from skyeye.utils.torch_utils import capture_graph
    from skyeye.distributed import BoxExchange

    def timed_leg(model_name, precision, size, steps, warmup):
        """Build + calibrate a detector, capture forward + NMS (+ the RCCL all-gather when world > 1) and time `steps` steps between
        barriers; returns everything the report needs."""
        model, P = build_model(model_name, precision, dev)
        B, S = a.batch, size
        frames_np = np.random.default_rng(rank).integers(0, 256, size=(B, 3, S, S), dtype=np.uint8)
        x = torch.from_numpy(frames_np).to(dev)
        if precision == "fp8":
            model.calibrate(x[:min(B, 16)])                      # per-tensor activation scales from frames of the workload
        calibrate_objectness(model, x, 0.01, a.conf)
        model.reuse_output_buffers(True)

        if a.slices:
            model.parallel_slices([int(v) for v in a.slices.split(",")])
        elif a.streams > 1:
            # the batch as equal slices on parallel HIP streams, one plan each (fp8: the slices share the calibration frames given above)
            model.parallel_slices(a.streams)

        # N > 1: the NMS kernel writes its rows / counts straight into the block the all-gather sends (skyeye.distributed.BoxExchange:
        # buffers allocated once per leg, no packing kernels, no allocation per step); three blocks = strict order + the two parities
        # of the pipelined loop.  N == 1: the module's own buffers.
        ex = [BoxExchange(B, 300, 7, dev, always_collective=force_ex) for _ in range(3)] if multi else None
        ag = {"ms": 0.0, "n": 0, "ev": []}

        # the collective runs on a SIDE stream behind the step that filled its block, beside the next step's kernels: a rank that arrives late at
        # the all-gather then delays the collective, not the other ranks' next forward pass.  A block is filled again two steps later (the pipelined
        # loop alternates two blocks): that step waits for the block's last collective first (SKY_BENCH_GATHER_INLINE=1: on the step's own stream).
        side = torch.cuda.Stream(device=dev) if multi and not os.environ.get("SKY_BENCH_GATHER_INLINE") else None
        sent = {}                                                                   # exchange block -> event behind its last collective

        def gather(e):
            """the RCCL all-gather of one block, bracketed by events on the stream it is enqueued on (allgather_ms of the line)"""
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if side is None:
                e0.record()
                res = e.gather()
                e1.record()
            else:
                filled = torch.cuda.Event()
                filled.record()                                                     # behind the replay that wrote the block
                with torch.cuda.stream(side):
                    side.wait_event(filled)
                    e0.record(side)
                    res = e.gather()
                    e1.record(side)
                sent[id(e)] = e1
            ag["ev"].append((e0, e1))
            return res

        def before_refill(e):
            """the step about to run writes block e again: behind the block's last collective"""
            ev = sent.pop(id(e), None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)

        def local_step():
            out = (ex[2].rows, ex[2].counts) if ex else None
            if a.with_raw or a.streams <= 1:
                det, _raw = model(x, return_raw=a.with_raw)                         # detector.py:300-324
                return nms_raw(det, a.conf, a.iou, max_detections=300,              # metrics.py:361-457, no host sync
                               **(dict(out=out[0], counts=out[1]) if out else {}))
            return model.detect_nms(x, a.conf, a.iou, max_detections=300, out=out)  # the same pair; each slice's NMS on its own stream

        graph = None
        if not a.no_graph:
            graph, held = capture_graph(local_step, warmup=2)                       # static shapes: one hipGraph replay per step

        def sync_step():                                                            # one batch, forward then its own NMS (the latency figure)
            if multi:
                before_refill(ex[2])
            if graph is not None:
                graph.replay()
                rows, counts = held
            else:
                rows, counts = local_step()
            if multi:
                rows, counts = gather(ex[2])                                        # RCCL over xGMI
            return rows, counts

        pipelined = not a.no_pipeline and not a.with_raw
        if pipelined:
            # throughput form: step k = forward(batch k) beside NMS(batch k - 1) (two buffer sets, two graphs replayed alternately);
            # finish() = the last batch's NMS, inside the timed region: K steps do K forward passes and K + 1 NMS calls
            def pipe_step(p, blk=None):
                # the boxes of the PREVIOUS batch (parity 1 - p) land in exchange block 1 - p (or in the block the caller names)
                if blk is None and ex:
                    blk = ex[1 - p]
                return model.detect_nms_pipelined(x, a.conf, a.iou, max_detections=300, parity=p,
                                                  out=(blk.rows, blk.counts) if blk is not None else None)
            graphs = None if a.no_graph else [capture_graph(lambda p=p: pipe_step(p), warmup=2) for p in (0, 1)]
            # two steps (parity 0 then 1) as ONE replay where nothing happens between them on the host: one graph-launch gap per two batches
            pair = None
            npair = max(1, int(os.environ.get("SKY_BENCH_PAIRS", "2")))          # even / odd pairs per replay
            if graphs is not None and not multi and not os.environ.get("SKY_BENCH_NO_PAIR"):
                pair = capture_graph(lambda: tuple(pipe_step(i & 1) for i in range(2 * npair)), warmup=1)
            # N > 1: the same replays; every step of a replay writes its OWN exchange block, the collectives of a replay's blocks run behind it on
            # the side stream, and two such graphs alternate so that a replay never waits for the collectives of the one before it
            mpairs = None
            if graphs is not None and multi and side is not None and not os.environ.get("SKY_BENCH_NO_PAIR"):
                mpairs = []
                for _g in range(2):
                    blks = [BoxExchange(B, 300, 7, dev, always_collective=force_ex) for _ in range(2 * npair)]
                    mpairs.append((capture_graph(lambda blks=blks: tuple(pipe_step(i & 1, blks[i]) for i in range(2 * npair)), warmup=1), blks))
            state = {"k": 0, "g": 0}

            def step():
                p = state["k"] & 1
                state["k"] += 1
                if multi:
                    before_refill(ex[1 - p])                                        # this step's NMS (of the previous batch) lands in block 1 - p
                if graphs is not None:
                    graphs[p][0].replay()
                    res = graphs[p][1]
                else:
                    res = pipe_step(p)
                if multi and res is not None:
                    res = gather(ex[1 - p])
                return res

            def run_steps(n):
                """n steps; pairs of (even, odd) steps as one replay where possible.  Returns the last step's result (the batch before it)."""
                res = None
                while n > 0:
                    if pair is not None and n >= 2 * npair and not (state["k"] & 1):
                        pair[0].replay()
                        res = pair[1][-1]
                        state["k"] += 2 * npair
                        n -= 2 * npair
                    elif mpairs is not None and n >= 2 * npair and not (state["k"] & 1):
                        (gr, _held), blks = mpairs[state["g"] & 1]
                        state["g"] += 1
                        for b in blks:
                            before_refill(b)
                        gr.replay()
                        for b in blks:
                            res = gather(b)                                         # one collective per step, in step order
                        state["k"] += 2 * npair
                        n -= 2 * npair
                    else:
                        res = step()
                        n -= 1
                return res

            def finish():
                res = model.detect_nms_flush(parity=(state["k"] - 1) & 1)
                if multi:
                    before_refill(ex[2])
                    ex[2].rows.copy_(res[0]); ex[2].counts.copy_(res[1])            # (the flush outside the loop: once per leg)
                    res = gather(ex[2])
                return res
        else:
            step = sync_step
            pair = None
            mpairs = None
            npair = 1

            def finish():
                return None

            def run_steps(n):
                res = None
                for _ in range(n):
                    res = step()
                return res

        run_steps(warmup)
        ag["ev"].clear()
        fence()
        t0 = time.perf_counter()
        res = run_steps(steps)
        last = finish()
        rows, counts = last if last is not None else res
        fence()
        dt = time.perf_counter() - t0
        dt_rank = dt
        ag_ms = sum(e0.elapsed_time(e1) for e0, e1 in ag["ev"]) / max(len(ag["ev"]), 1) if ag["ev"] else None
        ag["ev"].clear()
        rank_fps = None
        if multi:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            every = torch.zeros(world, device=dev, dtype=torch.float64)
            every[rank] = B * steps / dt_rank
            dist.all_reduce(every)
            rank_fps = [round(float(v), 1) for v in every.tolist()]
        # the strict order (a batch's NMS behind its own forward pass, one replay per batch): the like-for-like successor of the
        # figure rounds 1 - 2 reported, same number of steps, same fences
        strict = None
        if pipelined:
            for _ in range(2):
                sync_step()
            fence()
            t1 = time.perf_counter()
            for _ in range(steps):
                sync_step()
            fence()
            sdt = time.perf_counter() - t1
            if multi:
                t = torch.tensor([sdt], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                sdt = float(t.item())
            strict = {"frames_per_s": round(world * B * steps / sdt, 2), "ms_per_step": round(sdt / steps * 1e3, 3), "steps": steps,
                      "what": "forward then its own NMS" + (" then the all-gather" if world > 1 else "") + ", one graph replay per batch (--no-pipeline)"}
            ag["ev"].clear()
        return dict(model=model, P=P, x=x, frames_np=frames_np, step=sync_step, graph=graph, dt=dt, counts=counts, pipelined=pipelined,
                    steps_per_replay=2 * npair if (pipelined and (pair is not None or mpairs is not None)) else 1, strict=strict, allgather_ms=ag_ms, rank_fps=rank_fps)

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    leg = timed_leg(a.model, a.precision, a.size, a.steps, a.warmup)
    model, P, x, frames_np, step, graph, dt, counts = (leg[k] for k in ("model", "P", "x", "frames_np", "step", "graph", "dt", "counts"))
    B, S = a.batch, a.size
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

    # the three timing buckets the reference's CLIs print (validate.py:323-326): pre-process / inference / NMS, ms per image.
    # Pre-process there is .to(device).float() / 255 (validate.py:236-238): here the uint8 frames are already in HBM and the
    # conversion + /255 are fused into the stem's loader, so the bucket is empty by construction.
    def timed(fn, n=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n
    det_keep, _ = model(x, return_raw=a.with_raw)
    det_keep = det_keep.clone()
    inf_ms = timed(lambda: model(x, return_raw=a.with_raw))
    nms_ms = timed(lambda: nms_raw(det_keep, a.conf, a.iou, max_detections=300))
    # worst-case NMS leg: validate.py:117's conf 0.001 drives every image into the 30 000-candidate cap (metrics.py:393)
    worst_ms = timed(lambda: nms_raw(det_keep, 0.001, a.iou, max_detections=300), n=5)
    n_cand = int((det_keep[..., 4] > 0.001).sum(1).float().mean().item())

    frames = world * B * a.steps
    fps = frames / dt
    out = {
        "metric": "frames/sec @1280x1280 + p50 latency, skyeye_s 1-GPU and skyeye_l 8-GPU",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
        "p50_latency_ms": round(lat[len(lat) // 2], 3), "p99_latency_ms": round(lat[min(len(lat) - 1, int(0.99 * len(lat)))], 3),
        "rccl_ranks_seen": ranks_seen, "build": _native_build(),
        "strict_order": leg.get("strict"),
        "buckets_ms_per_image": {"pre_process": 0.0, "inference": round(inf_ms / B, 4), "nms": round(nms_ms / B, 4),
                                 "note": "validate.py:323-326 buckets; pre-process (uint8 -> float, /255) is fused into the stem's loader"},
        "nms_worst_case": {"conf": 0.001, "ms_per_batch": round(worst_ms, 3), "ms_per_image": round(worst_ms / B, 4),
                           "mean_candidates_per_image": n_cand, "cap": 30000},
        "config": {"workload": f"{a.model} {a.precision} batch={B}/GPU @{S}x{S}: uint8 frames in HBM -> backbone+neck+heads+"
                               f"decode -> NMS(conf {a.conf}, iou {a.iou}, max_det 300)"
                               + (" -> RCCL all-gather of boxes" if world > 1 else ""),
                   "global_batch": world * B, "frames_per_gpu": B, "image_size": S, "candidates_target": "1% > conf",
                   "mean_boxes_kept_per_image": round(kept_mean, 1), "parallelism": f"dp{world} (independent images)",
                   "hip_graph": graph is not None, "raw_levels_written": bool(a.with_raw),
                   "nms_one_batch_behind_forward": bool(leg.get("pipelined")), "steps_per_graph_replay": leg.get("steps_per_replay", 1),
                   "batch_slices_on_parallel_streams": a.streams},
    }
    if multi:
        rf = leg.get("rank_fps") or []
        out["multi_gpu"] = {"allgather_ms_per_step": None if leg.get("allgather_ms") is None else round(leg["allgather_ms"], 4),
                            "allgather_what": "hipEvent interval around the one fused all_gather_into_tensor of a step, mean over the timed steps, rank 0; "
                                              "the NMS kernel writes the exchanged block in place (no packing), buffers allocated once",
                            "per_rank_frames_per_s": rf, "per_rank_min": min(rf) if rf else None, "per_rank_max": max(rf) if rf else None}

    if rank == 0 and not a.no_roofline:
        from skyeye import _native as N
        h = model._engine([x])
        outs = [torch.empty(s, dtype=torch.float32, device=dev) for s in h.output_shapes()]
        stream = torch.cuda.current_stream(dev).cuda_stream
        prof = h.profile_forward([N.buffer_from_tensor(x)], [N.buffer_from_tensor(t) for t in outs], stream, iters=3)
        fam = {}
        last = "non_conv"
        frames_rd, chain_out = None, 0.0
        # an interval between two hipEvents holds one event's own processing besides the launch: the ops that launch nothing (computed by an
        # earlier kernel) measure exactly that (~5 us); it is taken off every interval so that a family's average agrees with the kernel
        # durations rocprofv3 reports for the same launches (profiles/*_kernel_stats_streams1.csv; what stays is the launch gap of an eager run)
        empty = [ms for ms, _fl, tag in prof if tag // 10000 in (0, 2) and tag % 10000 >= 9000]
        ev_over = min(empty) if empty else 0.0
        for i, (ms, fl, tag) in enumerate(prof):
            folded = tag // 10000 in (0, 2) and tag % 10000 >= 9000  # an op another launch computed (a convolution, the skipped import): its FLOPs belong to that launch
            rd, wr = h.op_io_bytes(i, with_raw=a.with_raw)
            info = h.op_info(i).split()
            if folded and info[0] == "import":
                frames_rd = rd                                   # the import launches nothing: the stem reads the caller's frames itself
                continue
            name = last if folded else family_of(tag)
            if folded and "head" in info and "cv3_head" in fam:
                name = "cv3_head"                                # the level the cv3 + level kernel computed (its op sits behind the neck's other launches)
            e = fam.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, variants=set()))
            ms = 0.0 if folded else max(ms - ev_over, 0.0)
            e["ms"] += ms; e["flops"] += fl
            if not folded:
                if frames_rd is not None:
                    rd, frames_rd = frames_rd, None              # (instead of the tensor the import would have written)
                e["launches"] += 1; e["bytes"] += rd + wr
                if tag // 10000 == 2:
                    e["variants"].add(tag % 10000)
                last = name
                chain_out = wr                                   # what this launch writes, until a folded op says its output stays on chip
            elif "head" in info:
                e["bytes"] += wr                                 # a detection level computed by the previous launch: its rows are written as well
            else:
                e["bytes"] += wr - chain_out                     # a folded convolution: the chain writes ITS output instead of the previous op's
                chain_out = wr
        peak = PEAK_TFLOPS[a.precision]
        pmc, pmc_src, pmc_lib = {}, None, None
        if a.model == "skyeye_s" and B == 32 and S == 1280 and a.precision == "bf16":
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json"))):     # latest PMC pass (tools/pmc_traffic.sh)
                try:
                    pj = json.load(open(f))
                    pmc, pmc_src, pmc_lib = pj["by_variant"], os.path.relpath(f, ROOT), pj.get("library_sources")
                except Exception:  # noqa: BLE001
                    pass
        pmc_fresh = bool(pmc_src) and pmc_lib == N.build_info()
        if not pmc_fresh:
            pmc = {}
        table = {}
        for name, e in fam.items():
            t = e["ms"] * 1e-3
            pb = [pmc[str(v)]["hbm_bytes_per_launch"] * 1.0 for v in e["variants"] if str(v) in pmc]
            table[name] = {"launches": e["launches"], "ms": round(e["ms"], 4), "avg_launch_ms": round(e["ms"] / e["launches"], 4),
                           "tflops": round(e["flops"] / t / 1e12, 1) if t else 0.0, "frac_mfma": round(e["flops"] / t / 1e12 / peak, 4) if t else 0.0,
                           "algorithmic_gbps": round(e["bytes"] / t / 1e9, 1) if t else 0.0,
                           "frac_hbm": round(e["bytes"] / t / 1e9 / PEAK_HBM_GBPS, 4) if t else 0.0,
                           "algorithmic_bytes_per_launch": round(e["bytes"] / e["launches"]),
                           "pmc_bytes_per_launch": round(sum(pb) / len(pb)) if pb else None}
            if table[name]["frac_hbm"] > 0.79:
                # more algorithmic bytes per second than HBM delivers (6.29 TB/s measured = 0.79 of spec): the working set of these
                # launches (CBAM / SPP re-read a map the previous launch just wrote) is served by the 256 MB Infinity Cache
                table[name]["served_from_cache"] = True
                table[name]["frac_hbm_note"] = "algorithmic bytes / time; not an HBM rate: the maps are re-read from the Infinity Cache"
        desc = {n: k for n, k, _ in FAMILIES}
        conv_names = [n for n in table if n not in ("non_conv",)]
        dom = max(conv_names, key=lambda n: table[n]["ms"])
        graph_flops = sum(e["flops"] for e in fam.values())
        graph_bytes = sum(e["bytes"] for e in fam.values())
        tot_ms = sum(e["ms"] for e in fam.values())
        e2e_tflops = fps / world * graph_flops / B / 1e12
        out["roofline"] = {
            "bound": "mfma", "achieved": round(e2e_tflops, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(e2e_tflops / peak, 4),
            "definition": "end-to-end: frames/s per GPU x algorithmic GFLOP/frame (2*MAC over conv/linear) / dense MFMA peak",
            "traffic": (round(sum(v["pmc_bytes_per_launch"] * v["launches"] for v in table.values() if v["pmc_bytes_per_launch"])) or None) if pmc_fresh else None,
            "traffic_source": pmc_src, "traffic_note": "PMC bytes per step summed over the convolution families (FETCH_SIZE x2 + WRITE_SIZE)",
            # the PMC pass is a separate, committed run: stale as soon as a kernel changes -- then it is an ERROR of this line, not a number
            "traffic_measured_on_this_build": pmc_fresh if pmc_src else None,
            "traffic_error": None if (pmc_fresh or not pmc_src) else (f"{pmc_src} was measured on library sources {pmc_lib}, the loaded library is "
                                                                       f"{N.build_info()}: traffic and pmc_bytes_per_launch withheld (re-run tools/pmc_traffic.sh)"),
            "graph_gflop_per_frame": round(graph_flops / B / 1e9, 2), "baseline_gflop_per_frame": GFLOP_PER_FRAME.get((a.model, S)),
            "algorithmic_bytes_per_step": round(graph_bytes), "end_to_end_hbm_gbps": round(fps / world / B * graph_bytes / 1e9, 1),
            "end_to_end_frac_hbm": round(fps / world / B * graph_bytes / 1e9 / PEAK_HBM_GBPS, 4),
            "graph_ms_per_step": round(tot_ms, 3), "graph_launches": len(prof), "event_interval_overhead_ms": round(ev_over, 4),
            "dominant": dict(family=dom, kernel=desc.get(dom, dom), **table[dom]),
            "families": table,
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
        # second comparator (optional): the same graph in torch.nn.functional on the host cores (tests/torch_graph.py:
        # checker code, plain variants only), one frame, bounded to a few seconds
        torch_cmp = None
        if not okw["enhanced"] and not okw["head_attention"]:
            try:
                import torch_graph as TG
                nt = torch.get_num_threads()
                Pt = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in Pc.items() if np.asarray(v).dtype != np.int64}
                xt = torch.from_numpy(frames_np[:1].astype(np.float32) / np.float32(255.0))
                TG.detector_forward(Pt, xt, nc)
                t1 = time.perf_counter()
                reps = 0
                while reps < 4 and time.perf_counter() - t1 < 6.0:
                    TG.detector_forward(Pt, xt, nc)
                    reps += 1
                torch_cmp = {"value": round(reps / (time.perf_counter() - t1), 4), "unit": "frames/s", "threads": nt,
                             "what": "the same graph in torch.nn.functional on the CPU (forward + decode, no NMS), the build's own code"}
            except Exception as ex:  # noqa: BLE001
                torch_cmp = {"error": str(ex)[:200]}
        out["cpu_baseline"] = {"value": round(n / cpu_dt, 4), "unit": "frames/s", "cores": O.threads(), "kind": "port", "torch_graph": torch_cmp,
                               "sample": f"{n} frame(s) of the same workload ({a.model} fp32 @{S}x{S}, forward+decode+NMS), "
                                         f"oracle/ C+OpenMP port of the reference path, {O.threads()} threads used of {os.cpu_count()} host "
                                         "CPUs visible; the reference's own PyTorch-CPU path measured ~1.09 frames/s on 8 cores at "
                                         "survey time (BASELINE.md section 2) -- a reported baseline, not the target"}

    # ---- the other configurations of BASELINE.json, 5-step legs (never part of `value`) ----
    default_headline = a.model == "skyeye_s" and a.precision == "bf16" and S == 1280
    if default_headline and not a.no_other_configs:
        del leg, model, step, graph
        torch.cuda.empty_cache()
        legs = OTHER_CONFIGS if world == 1 else [("skyeye_l", "bf16", 1280)]      # N > 1: the model the metric names at 8 GPUs
        others = []
        for name, prec, size in legs:
            try:
                lg = timed_leg(name, prec, size, 5, 2)
                f = world * B * 5 / lg["dt"]
                gf = plan_gflop_per_frame(lg["model"], lg["x"])                    # the plan's own count: includes the attention FLOPs of skyeye_s_ha
                entry = {"workload": f"{name} {prec} batch={B}/GPU @{size}x{size}, forward + NMS" + (" + RCCL all-gather" if world > 1 else ""),
                         "dtype": prec, "frames_per_s": round(f, 1), "ms_per_step": round(lg["dt"] / 5 * 1e3, 3), "steps": 5,
                         "graph_gflop_per_frame": round(gf, 2), "peak_tflops": PEAK_TFLOPS[prec],
                         "roofline_frac": round(f / world * gf * 1e9 / (PEAK_TFLOPS[prec] * 1e12), 4),
                         "strict_order_frames_per_s": (lg.get("strict") or {}).get("frames_per_s"),
                         "batch_slices_on_parallel_streams": a.streams}
                if prec == "fp32":
                    entry["what"] = "fp32_engine: the exact-mode engine (v_mfma_f32_16x16x4_f32; the one the 1e-4 parity tests pin) on the headline workload"
                if multi:
                    entry["allgather_ms_per_step"] = None if lg.get("allgather_ms") is None else round(lg["allgather_ms"], 4)
                    entry["per_rank_frames_per_s"] = lg.get("rank_fps")
                others.append(entry)
                del lg
                torch.cuda.empty_cache()
            except Exception as ex:  # noqa: BLE001 -- a failing side leg must not lose the headline line
                if world > 1:
                    raise                       # (a rank that skipped a leg would leave the others waiting in its collectives)
                others.append({"workload": f"{name} {prec} @{size}", "error": str(ex)[:300]})
        out["other_configs"] = others
    if world > 1:
        out["config"]["scaling_curve_model"] = (f"the N = 1 .. 8 values of this line are {a.model} (BASELINE.json configs[1] replicated per GPU, weak scaling); "
                                                "the skyeye_l leg the metric names at 8 GPUs is other_configs[0]")
    if rank == 0:
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
