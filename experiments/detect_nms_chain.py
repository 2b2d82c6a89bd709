"""Round-3 experiment, reconstructed in round 4 from experiments/README.md ("Join-free chain of steps") and the crash log
(gpurun_out/t_sl.log: Segmentation fault in torch/cuda/graphs.py capture_end <- capture_graph <- test_detect_nms_chain_...): the source
was never committed.  NOT part of the package; kept so that the failure has a source.

Idea: n pipelined steps as ONE enqueue without a fork / join bubble between them -- slice i of batch k + 1 follows slice i of batch k
on its own stream, the NMS of batch k (its own stream) waits for both slices of batch k, a forward pass that re-uses a detection
buffer waits for the NMS that read it.  All dependencies run directly between the slice streams and the NMS stream.

Why `hipStreamEndCapture` faulted (worked out on the CPU with skyeye.utils.torch_utils.CaptureLedger, tests/test_capture_ledger.py):
with ``join=False`` (the round-3 form) the slice streams and the NMS stream enter the capture -- each waits for an event recorded in
the capturing caller's stream -- and NOTHING orders their tails before the caller's stream again.  That is an invalid capture
(CUDA: cudaErrorStreamCaptureUnjoined).  Keeping the event objects alive does not change the topology, which is why "wait_stream
pairs and explicit events kept alive alike" failed.  The ROCm 7.0 runtime of this image dereferences the unjoined capture in
hipStreamEndCapture instead of returning the error.  ``join=True`` adds the one missing edge (caller waits for the NMS stream, whose
tail is behind every slice's last forward pass) and is a legal capture; ``capture_graph`` now refuses the ``join=False`` form with a
Python error before the runtime sees it (tests/test_gpu_capture_guard.py).

Round 4, second finding (experiments/stagger_probe.py): with THREE or more batches this function makes the slice streams wait for the NMS
stream (the forward pass of batch k re-uses the detection buffer the NMS of batch k - 2 read) while the NMS stream waits for the slice
streams -- mutual waits between two forked streams.  That DAG is legal and, with ``join=True``, joined; hipStreamEndCapture faults on it all
the same (8-step chain, faulthandler: torch/cuda/graphs.py capture_end).  With one detection buffer per step (no back edge) the same
8-step chain captures and replays.  ``CaptureLedger.wait`` flags the back edge before it is made and ``capture_graph`` refuses it."""
import torch


def detect_nms_chain(model, batches, conf=0.25, iou=0.45, max_detections=300, join=False):
    """-> [(rows, counts)] per batch.  ``model``: a SkyEyeDetector with ``parallel_slices(2)`` semantics done by hand; ``batches``: a
    list of input tensors of one geometry (captured: the same storage every replay)."""
    from skyeye import _native as N
    from skyeye.utils.metrics import nms_raw

    x0 = model._prepare_input(batches[0])
    dev, B = x0.device, x0.shape[0]
    nsl, half = 2, x0.shape[0] // 2
    st = model.__dict__.setdefault("_chain", {})
    if "streams" not in st:
        st["streams"] = [torch.cuda.Stream(device=dev) for _ in range(nsl)]
        st["nms"] = torch.cuda.Stream(device=dev)
    slices, nms_s = st["streams"], st["nms"]
    cur = torch.cuda.current_stream(dev)
    ents = [model._engine_entry([x0[i * half:(i + 1) * half]], None, slot=i + 1)[1] for i in range(nsl)]
    shapes = ents[0].output_shapes()
    if st.get("B") != (B, max_detections, len(batches)):
        st["B"] = (B, max_detections, len(batches))
        st["det"] = [torch.empty((B,) + tuple(shapes[0][1:]), dtype=torch.float32, device=dev) for _ in range(2)]
        st["out"] = [(torch.empty((B, max_detections, 7), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev))
                     for _ in range(len(batches))]
    for s_ in slices:
        s_.wait_stream(cur)                                   # fork: the slice streams enter the capture
    results = []
    for k, xb in enumerate(batches):
        xk = model._prepare_input(xb)
        det = st["det"][k & 1]
        for i, s_ in enumerate(slices):
            if k >= 2:
                s_.wait_stream(nms_s)                         # the NMS of batch k - 2 read this detection buffer
            with torch.cuda.stream(s_):
                outs = [N.buffer_from_tensor(det[i * half:(i + 1) * half])] + [N.null_buffer()] * (len(shapes) - 1)
                ents[i].forward([N.buffer_from_tensor(xk[i * half:(i + 1) * half])], outs, s_.cuda_stream)
        for s_ in slices:
            nms_s.wait_stream(s_)                             # the NMS stream enters the capture here (k = 0)
        with torch.cuda.stream(nms_s):
            nms_raw(det, conf, iou, max_detections=max_detections, out=st["out"][k][0], counts=st["out"][k][1])
        results.append(st["out"][k])
    if join:
        cur.wait_stream(nms_s)                                # the one edge round 3 left out: every tail is ordered before the caller again
    return results
