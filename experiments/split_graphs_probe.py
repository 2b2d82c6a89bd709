#!/usr/bin/env python3
"""Experiment (round 4): instead of ONE hipGraph with two slice branches + the NMS branch (fork / join edges inside the graph), every branch
is its OWN linear graph launched on its own stream, ordered by eager stream events: does the runtime run two linear graphs on two streams
more concurrently than two branches of one graph?  (experiments/stagger_probe.py: an 8-step three-branch chain graph lost 12 % against the
per-step fork / join graph.)  Same box, same process, skyeye_s bf16 B = 32 @1280."""
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

dev = torch.device("cuda", 0)
model, _ = build_model("skyeye_s", "bf16", dev)
B, S, STEPS = 32, 1280, 64
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, size=(B, 3, S, S), dtype=np.uint8)).to(dev)
calibrate_objectness(model, x, 0.01, 0.25)
model.reuse_output_buffers(True).parallel_slices(2)
half = B // 2


def timed(step, n=STEPS):
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)


pair = capture_graph(lambda: tuple(model.detect_nms_pipelined(x, 0.25, 0.45, max_detections=300, parity=i & 1) for i in range(4)), warmup=2)
print(f"shipped: one graph, 4 steps per replay      {timed(pair[0].replay, STEPS // 4) * 4:9.1f} frames/s", flush=True)
want = tuple(t.clone() for t in model.detect_nms(x, 0.25, 0.45, max_detections=300))

# ---- one linear graph per (slice, parity) and per NMS parity, launched on three streams
ents = [model._engine_entry([x[i * half:(i + 1) * half]], None, slot=i + 1)[1] for i in range(2)]
shapes = ents[0].output_shapes()
det = [torch.empty((B,) + tuple(shapes[0][1:]), dtype=torch.float32, device=dev) for _ in range(2)]
outs = [(torch.empty((B, 300, 7), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev)) for _ in range(2)]


def fwd(i, p):
    o = [N.buffer_from_tensor(det[p][i * half:(i + 1) * half])] + [N.null_buffer()] * (len(shapes) - 1)
    ents[i].forward([N.buffer_from_tensor(x[i * half:(i + 1) * half])], o, torch.cuda.current_stream(dev).cuda_stream)


g_fwd = [[capture_graph(lambda i=i, p=p: fwd(i, p), warmup=1)[0] for p in range(2)] for i in range(2)]
g_nms = [capture_graph(lambda p=p: nms_raw(det[p], 0.25, 0.45, max_detections=300, out=outs[p][0], counts=outs[p][1]), warmup=1)[0] for p in range(2)]
sl = [torch.cuda.Stream(device=dev) for _ in range(2)]
sn = torch.cuda.Stream(device=dev)
state = {"k": 0}
ev_nms = [None, None]


def split_step():
    """step k: forward(batch k) of both slices (own streams) beside NMS(batch k - 1) (own stream)"""
    k = state["k"]
    p = k & 1
    for i in range(2):
        if ev_nms[p] is not None:
            sl[i].wait_event(ev_nms[p])                     # the NMS of batch k - 2 read det[p]
        with torch.cuda.stream(sl[i]):
            g_fwd[i][p].replay()
    if k > 0:
        q = 1 - p
        with torch.cuda.stream(sn):
            g_nms[q].replay()                              # (sn waited for the slices of batch k - 1 at the end of the previous step)
            ev_nms[q] = sn.record_event()
    for i in range(2):
        sn.wait_stream(sl[i])                              # NMS(batch k) may start when both slices of batch k are done
    state["k"] = k + 1


print(f"split: one linear graph per branch          {timed(split_step):9.1f} frames/s", flush=True)
torch.cuda.synchronize()
with torch.cuda.stream(sn):
    g_nms[(state["k"] - 1) & 1].replay()
torch.cuda.synchronize()
r, c = outs[(state["k"] - 1) & 1]
print("results equal to detect_nms:", bool(torch.equal(r, want[0]) and torch.equal(c, want[1])))
