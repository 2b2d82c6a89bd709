"""time_sync / select_device / scale_img of reference skyeye/utils/torch_utils.py (:70-118 the timing convention of the CLIs,
:262-288 the resize of test-time augmentation)."""
import ctypes
import math
import time

import torch

from .. import _native as N


def time_sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()


def select_device(device=""):
    if str(device).lower() == "cpu" or not torch.cuda.is_available():
        raise RuntimeError("the SkyEye HIP engine needs an MI355X: there is no CPU path (reference select_device, torch_utils.py:70-106)")
    idx = 0 if device in ("", "cuda") else int(str(device).replace("cuda:", "").split(",")[0])
    return torch.device("cuda", idx)


def scale_img_geometry(h, w, ratio, same_shape=False, gs=32):
    """-> (resized (h, w), padded (h, w)): torch_utils.py:275-288 (python float arithmetic, ``int()`` truncation, ``math.ceil``)."""
    s = (int(h * ratio), int(w * ratio))
    if not same_shape:
        h, w = (math.ceil(x * ratio / gs) * gs for x in (h, w))
    return s, (h, w)


def scale_img(img, ratio=1.0, same_shape=False, gs=32, flip=None):
    """reference torch_utils.py:262-288 on the device: bilinear resize (align_corners=False) of ``img`` [B, C, H, W] to
    ``int(h * ratio) x int(w * ratio)``, padded bottom / right with 0.447 to a multiple of ``gs`` (or back to (h, w) with
    ``same_shape``).  ``flip`` (None, 2 or 3) folds the caller's ``img.flip(flip)`` into the same pass.  A uint8 image is
    divided by 255 first (validate.py:236-238); the result is always float32.  ``ratio == 1.0`` without flip returns ``img``."""
    from .metrics import _handle
    if not (torch.is_tensor(img) and img.is_cuda and img.dim() == 4):
        raise N.SkyEyeNativeError("scale_img: img must be a [B, C, H, W] tensor on the HIP device (no CPU path)")
    flip = int(flip or 0)
    if ratio == 1.0 and not flip:
        return img
    if img.dtype != torch.uint8:
        img = img.float()
    src = img.contiguous()
    B, C, H, W = src.shape
    s, p = ((H, W), (H, W)) if ratio == 1.0 else scale_img_geometry(H, W, ratio, same_shape, gs)
    out = torch.empty((B, C, p[0], p[1]), dtype=torch.float32, device=img.device)
    h = _handle(img.device.index or 0)
    stream = torch.cuda.current_stream(img.device).cuda_stream
    N.check(h.L.sky_scale_img(h.h, src.data_ptr(), N.SKY_IO_U8 if src.dtype == torch.uint8 else N.SKY_IO_F32, B, C, H, W, out.data_ptr(),
                              s[0], s[1], p[0], p[1], flip, 0.447, ctypes.c_void_p(stream)), h.h)
    return out


class CaptureLedger:
    """Book of the cross-stream dependencies made while a hipGraph is being captured (pure Python: streams are any hashable keys,
    so the rules are testable without a device -- tests/test_capture_ledger.py).

    A stream capture is a fork / join DAG rooted in the ORIGIN stream (the one ``hipStreamBeginCapture`` was called on).  Another
    stream joins the capture by waiting for an event recorded in a capturing stream, and before ``hipStreamEndCapture`` the tail of
    every such stream must be ordered before the origin's tail again; otherwise the capture is invalid (``StreamCaptureUnjoined``).
    The ROCm 7.0 runtime of this image does not return that error: ``hipStreamEndCapture`` faults (round 3: ``detect_nms_chain``,
    experiments/detect_nms_chain.py -- slice and NMS streams ordered among themselves, never back into the origin).  The ledger
    finds the unjoined streams BEFORE the capture ends, so ``capture_graph`` can join them, end the capture legally and raise.

    A second shape faults in ``hipStreamEndCapture`` although it is a legal DAG and fully joined (round 4, experiments/stagger_probe.py:
    an 8-step chain whose slice streams wait for the NMS stream's events while the NMS stream waits for theirs): MUTUAL waits between two
    forked streams.  With one detection buffer per step -- no slice stream ever waits for the NMS stream -- the same chain captures and
    replays.  ``wait`` returns a message for such an edge BEFORE it is made, so that the caller can refuse it.

    What it is told: ``record(s)`` -> event token (the tail of stream s at this moment), ``wait(s, token)``.  Kernel launches are not
    seen; every record or wait on a stream counts as new activity on it, which is what a launch between them would be."""

    def __init__(self, origin):
        self.origin = origin
        self.seq = {origin: 0}            # activity counter per stream
        self.cover = {origin: {}}         # stream -> {other stream: highest activity of it ordered before this stream's tail}
        self.captured = {origin}
        self.problems = []

    def record(self, s):
        self.seq[s] = self.seq.get(s, 0) + 1
        return (s, self.seq[s], dict(self.cover.get(s, {})), s in self.captured)

    def wait(self, s, token):
        src, n, cov, src_captured = token
        if not src_captured:
            if s in self.captured:
                self.problems.append(f"capturing stream {s!r} waits for an event recorded outside the capture on stream {src!r} "
                                     "(a dependency across the capture boundary: StreamCaptureIsolation)")
            return
        fatal = None
        if s != self.origin and src != self.origin and self.cover.get(src, {}).get(s, 0) > 0:
            fatal = (f"forked stream {s!r} waits for forked stream {src!r}, which has itself waited for {s!r}: mutual waits between two forked "
                     "streams make this runtime's hipStreamEndCapture fault (a legal DAG; order such work through the capturing stream, or give "
                     "every step its own buffers so that the back edge is not needed)")
            self.problems.append(fatal)
        self.captured.add(s)
        self.seq[s] = self.seq.get(s, 0) + 1
        c = self.cover.setdefault(s, {})
        for k, v in cov.items():
            if c.get(k, 0) < v:
                c[k] = v
        if c.get(src, 0) < n:
            c[src] = n
        return fatal

    def unjoined(self):
        """Streams that joined the capture and whose tail is not ordered before the origin's tail."""
        c = self.cover.get(self.origin, {})
        return [s for s in self.captured if s != self.origin and c.get(s, 0) < self.seq.get(s, 0)]


class _LedgerPatch:
    """Routes torch's event calls through a CaptureLedger for the duration of a capture.  ``Stream.wait_stream``, ``record_event``
    and ``wait_event`` all end in ``Event.record(stream)`` / ``Event.wait(stream)`` (torch/cuda/streams.py), so those two are the
    whole surface.  The event objects are kept until the capture has ended (``wait_stream`` drops its temporary event at once)."""

    def __init__(self, ledger):
        self.ledger, self.tokens, self.streams, self.saved = ledger, {}, {}, None

    def __enter__(self):
        L, tokens, streams = self.ledger, self.tokens, self.streams
        E = torch.cuda.Event
        rec, wai = self.saved = (E.record, E.wait)

        def ev_record(self_, stream=None):
            st = stream if stream is not None else torch.cuda.current_stream()
            streams[st.cuda_stream] = st
            tokens[id(self_)] = (self_, L.record(st.cuda_stream))
            return rec(self_, st)

        def ev_wait(self_, stream=None):
            st = stream if stream is not None else torch.cuda.current_stream()
            streams[st.cuda_stream] = st
            if id(self_) in tokens:
                fatal = L.wait(st.cuda_stream, tokens[id(self_)][1])
                if fatal:                                 # refused BEFORE the edge exists: the capture can still end legally
                    raise N.SkyEyeNativeError("capture_graph: " + fatal)
            return wai(self_, st)

        E.record, E.wait = ev_record, ev_wait
        return self

    def __exit__(self, *exc):
        torch.cuda.Event.record, torch.cuda.Event.wait = self.saved
        return False


def capture_graph(fn, warmup=2):
    """Capture ``fn()`` -- a static-shape chain of engine calls such as forward + ``nms_raw``, no host synchronisation inside --
    into ONE hipGraph: ``graph, outputs = capture_graph(step); graph.replay()`` re-runs all its launches (77 for skyeye_s plus
    NMS) with a single host call and refreshes ``outputs`` in place.  The warm-up runs on the capture stream first, so that every
    one-time allocation (plans, NMS workspace, LDS attributes) happens before the capture starts.  Inputs captured by ``fn``
    must keep their storage; to feed new frames copy them into the captured input tensor.

    ``fn`` may fork work onto other streams (``parallel_slices``, ``detect_nms_pipelined``); every such stream must be joined back
    into the calling stream before ``fn`` returns.  A ``CaptureLedger`` follows the stream waits during the capture: streams left
    unjoined are joined here, the capture is ended legally, the graph is dropped and ``SkyEyeNativeError`` names them -- this
    runtime's ``hipStreamEndCapture`` faults on an unjoined capture instead of returning ``hipErrorStreamCaptureUnjoined``."""
    if not torch.cuda.is_available():
        raise N.SkyEyeNativeError("capture_graph needs the HIP device (no CPU path)")
    from . import metrics as _metrics
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    keep = []
    _metrics._KEEP.append(keep)
    problems = []
    try:
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        ledger = CaptureLedger(side.cuda_stream)
        with torch.cuda.graph(graph, stream=side):
            with _LedgerPatch(ledger) as patch:
                try:
                    outputs = fn()
                finally:
                    for h in ledger.unjoined():
                        problems.append(f"stream {h:#x} joined the capture and was not joined back into the capture stream")
                        ev = torch.cuda.Event()
                        patch.saved[0](ev, patch.streams[h])               # make the capture legal: it must END, valid or not
                        patch.saved[1](ev, side)
                        patch.tokens[id(ev)] = (ev, None)
                    problems.extend(ledger.problems)
    finally:
        _metrics._KEEP.pop()
    if problems:
        del graph
        raise N.SkyEyeNativeError("capture_graph: invalid capture topology (graph dropped): " + "; ".join(problems))
    # the graph replays into the NMS workspace of the utility handle(s) and the arenas of the plans it was captured with: they
    # live as long as the graph; the capture stream dies with this call, so its table entry goes (a new stream may get the same pointer)
    graph._sky_keep = keep
    _metrics.forget_stream(torch.cuda.current_device(), side.cuda_stream)
    return graph, outputs
