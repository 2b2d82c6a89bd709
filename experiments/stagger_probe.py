#!/usr/bin/env python3
"""Experiment (round 4): n pipelined steps as ONE captured chain in which batch slice 1 starts when slice 0 is `stagger_op` launches into its
forward pass (sky_forward_mark) and stays that far behind for the whole replay, so that the two slices run DIFFERENT layers at any time
(an HBM-bound 1x1 beside a matrix-bound 3x3) instead of the same kernel twice.  Topology: slice i of batch k + 1 follows slice i of batch k
on its own stream; the NMS of batch k (own stream) waits for both slices of batch k; a forward pass that re-uses a detection buffer waits for
the NMS that read it; ONE final join (experiments/detect_nms_chain.py has the history of this shape).  Prints frames/s per setting against
the shipped pair-graph loop of bench.py on the same box.

NEEDS experiments/sky_forward_mark.patch (a one-shot "record this event behind launch k of the next forward pass" entry point, applied to
engine.cpp / skyeye_hip.h / _native.py for the measurement and reverted: the result below does not justify an entry point).

Result (MI355X, skyeye_s bf16 B = 32 @1280, same process): shipped loop 7 580 - 7 600 frames/s; chain of 8 steps 6 668 without stagger,
6 658 / 6 675 with slice 1 starting behind launch 24 / 32 of slice 0; chain of 2 steps 6 423.  The join-free chain is 12 % SLOWER than forking and
joining the slices every step, and a stagger changes nothing: replayed from a hipGraph the three long branches are not scheduled the way three
eager streams would be.  First form of the chain (two detection buffers, the forward pass of batch k waits for the NMS of batch k - 2): segmentation
fault in hipStreamEndCapture for 8 steps -- mutual waits between forked streams, see experiments/detect_nms_chain.py."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skyeye-aerial-object-detection-using-yolo_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch

from bench import build_model, calibrate_objectness
from skyeye import _native as N
from skyeye.utils.metrics import nms_raw
from skyeye.utils.torch_utils import capture_graph

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--chains", default="8,16")
ap.add_argument("--staggers", default="0,16,24,32,40")
a = ap.parse_args()
dev = torch.device("cuda", 0)
model, _ = build_model("skyeye_s", "bf16", dev)
B, S = 32, 1280
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, size=(B, 3, S, S), dtype=np.uint8)).to(dev)
calibrate_objectness(model, x, 0.01, 0.25)
model.reuse_output_buffers(True)
half = B // 2
ents = [model._engine_entry([x[i * half:(i + 1) * half]], None, slot=i + 1)[1] for i in range(2)]
shapes = ents[0].output_shapes()
n_ops = ents[0].stats()["launches"]
slices = [torch.cuda.Stream(device=dev) for _ in range(2)]
nms_s = torch.cuda.Stream(device=dev)
NDET = 16                                                   # one detection buffer per step of a chain: no slice stream ever waits for the NMS stream
det = [torch.empty((B,) + tuple(shapes[0][1:]), dtype=torch.float32, device=dev) for _ in range(NDET)]


def chain(n, stagger_op, outs):
    cur = torch.cuda.current_stream(dev)
    mark = torch.cuda.Event()
    mark.record(cur)                                      # (creates the hipEvent; re-recorded by the engine inside slice 0's first pass)
    slices[0].wait_stream(cur)
    for k in range(n):
        d = det[k % NDET]
        for i, s_ in enumerate(slices):
            if k >= NDET:
                raise RuntimeError("chain longer than the detection buffers")
            if k == 0 and i == 1:
                if stagger_op > 0:
                    s_.wait_event(mark)                   # slice 1 starts behind launch `stagger_op` of slice 0
                else:
                    s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                if k == 0 and i == 0 and stagger_op > 0:
                    ents[0].forward_mark(stagger_op, mark.cuda_event)
                o = [N.buffer_from_tensor(d[i * half:(i + 1) * half])] + [N.null_buffer()] * (len(shapes) - 1)
                ents[i].forward([N.buffer_from_tensor(x[i * half:(i + 1) * half])], o, s_.cuda_stream)
        for s_ in slices:
            nms_s.wait_stream(s_)
        with torch.cuda.stream(nms_s):
            nms_raw(d, 0.25, 0.45, max_detections=300, out=outs[k][0], counts=outs[k][1])
    cur.wait_stream(nms_s)                                # the one join: every tail is ordered before the caller again
    return outs


def timed(replay, steps_per_replay):
    for _ in range(3):
        replay()
    torch.cuda.synchronize()
    reps = max(1, a.steps // steps_per_replay)
    t0 = time.perf_counter()
    for _ in range(reps):
        replay()
    torch.cuda.synchronize()
    return B * reps * steps_per_replay / (time.perf_counter() - t0)


# reference: the shipped loop (two slices forked / joined per step, NMS one batch behind, two even / odd pairs per replay)
model.parallel_slices(2)
pair = capture_graph(lambda: tuple(model.detect_nms_pipelined(x, 0.25, 0.45, max_detections=300, parity=i & 1) for i in range(4)), warmup=2)
print(f"shipped pair graph (4 steps per replay): {timed(pair[0].replay, 4):9.1f} frames/s", flush=True)
want = tuple(t.clone() for t in model.detect_nms(x, 0.25, 0.45, max_detections=300))
torch.cuda.synchronize()
print("reference result taken", flush=True)
for n in [int(v) for v in a.chains.split(",")]:
    outs = [(torch.empty((B, 300, 7), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev)) for _ in range(n)]
    for st in [int(v) for v in a.staggers.split(",")]:
        if st >= n_ops:
            continue
        print(f"capturing chain n = {n}, stagger {st}", flush=True)
        g, res = capture_graph(lambda: chain(n, st, outs), warmup=1)
        print("captured", flush=True)
        fps = timed(g.replay, n)
        ok = all(torch.equal(r, want[0]) and torch.equal(c, want[1]) for r, c in res)
        print(f"chain of {n:2d} steps, slice 1 behind launch {st:2d} of {n_ops}: {fps:9.1f} frames/s   results {'equal to detect_nms' if ok else 'DIFFER'}", flush=True)
        del g
