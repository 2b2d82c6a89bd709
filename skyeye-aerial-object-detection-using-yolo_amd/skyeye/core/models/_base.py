"""Common machinery of the drop-in modules: parameter holders + the bridge to libskyeye_hip.so.

Every class in ``blocks.py`` / ``attention.py`` / ``backbone.py`` / ``detector.py`` keeps the reference's
constructor signature, attribute names and ``state_dict()`` keys, but its ``forward`` is one call into the HIP
engine (``sky_forward``).  There is deliberately no PyTorch implementation of the math anywhere in this package:
without the native library or without a HIP device ``forward`` raises.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from ... import _native as N


# Weight staleness.  An engine packs (BN-folds, converts) the weights once; it must notice every later change, also the ones
# that never pass through the NativeModule that owns the engine: ``model.backbone.load_state_dict(sd)`` on a plain nn.Module
# child, ``p.data.copy_()``, an optimizer step.  Two cheap signals cover them:
#   * _EPOCH, a process-wide counter bumped whenever a tensor attribute is (re)bound or a module is moved / cast
#     (``_apply``): the cached list of a module's tensors is rebuilt when it moved;
#   * the sum of ``tensor._version`` over that list (in-place writes bump it), taken on every forward (~75 us for skyeye_s).
_EPOCH = [0]

# developer switches read by sky_plan (csrc/sky_kernels.h PlanOpt): part of the engine cache key, so that a test which
# toggles one between two calls of the same module gets a plan made under the new value
_PLAN_SWITCHES = ("SKY_CONV_HALO", "SKY_HALO_NF8", "SKY_HALO_S2", "SKY_NO_STREAM", "SKY_NO_RING", "SKY_STREAM_OLDGRID",
                  "SKY_NO_FUSED_IMPORT", "SKY_FUSE", "SKY_NO_FUSE", "SKY_NO_SPP_PYRAMID", "SKY_ATTN_VALU", "SKY_HALO_SKIP",
                  "SKY_NO_FUSE_CV1", "SKY_NO_STEM_DOWN", "SKY_SUBBATCH", "SKY_NO_WINATTN", "SKY_NO_CSP_STAGE", "SKY_NO_HEAD_STREAM", "SKY_HEAD_STREAM", "SKY_NO_BNECK128", "SKY_BNECK128", "SKY_NO_BNECK64W", "SKY_NO_DEEP3X3", "SKY_NO_IN2", "SKY_NO_CV3_HEAD", "SKY_NO_GEMM1X1", "SKY_GEMM1X1")


def _bump_epoch():
    _EPOCH[0] += 1


# ----------------------------------------------------------------------------- parameter holders
class _Holder(nn.Module):
    """A module that only owns tensors (same names / shapes as the torch.nn layer it stands for)."""

    def __setattr__(self, name, value):
        if isinstance(value, torch.Tensor):
            _bump_epoch()
        super().__setattr__(name, value)

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        _bump_epoch()
        return r

    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError(f"{type(self).__name__} only holds parameters; computation runs inside libskyeye_hip.so "
                           "through the enclosing SkyEye module")


class Conv2dParams(_Holder):
    """Stands for nn.Conv2d(cin, cout, k, ..., bias=bias) (reference blocks.py:31, detector.py:56-59)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = (kernel_size, kernel_size)
        self.stride, self.padding = (stride, stride), (padding, padding)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(in_channels * kernel_size * kernel_size)
            nn.init.uniform_(self.bias, -bound, bound)


class BatchNormParams(_Holder):
    """Stands for nn.BatchNorm2d(ch) in eval mode (eps 1e-5), reference blocks.py:32."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class LinearParams(_Holder):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(in_features)
            nn.init.uniform_(self.bias, -bound, bound)


class LayerNormParams(_Holder):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class Marker(_Holder):
    """Parameter-free placeholder (nn.SiLU / nn.Identity / nn.ReLU / pooling attributes of the reference)."""

    def __init__(self, name):
        super().__init__()
        self.kind = name

    def extra_repr(self):
        return self.kind


# ----------------------------------------------------------------------------- native bridge
class NativeModule(nn.Module):
    """Base of every drop-in module: forward() == sky_forward on a planned static graph."""

    _sky_module = None          # key of _native.MODULES

    def __init__(self):
        super().__init__()
        self.__dict__["_engines"] = {}      # (precision, device, shapes, switches) -> (Handle, weights fingerprint)
        self.__dict__["_weights_version"] = 0
        self.__dict__["_precision"] = None   # None: follow parameter dtype (fp32 -> exact, half/bf16 -> bf16)
        self.__dict__["_out_cache"] = None   # reuse_output_buffers(True): {engine key (NOT id(handle): ids are recycled): output tensors}

    def __setattr__(self, name, value):
        if isinstance(value, torch.Tensor):
            _bump_epoch()
        super().__setattr__(name, value)

    # -- configuration handed to sky_create; subclasses override
    def _sky_config(self):
        raise NotImplementedError

    # -- reference API surface that changes weights / precision
    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.refresh_weights()
        return r

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.refresh_weights()
        return r

    def refresh_weights(self):
        """Force engines to re-pack weights on the next forward.  Not needed after in-place edits of the parameters themselves
        (``p.mul_()``, ``p.copy_()`` under no_grad, an optimizer step) or ``load_state_dict`` on any sub-module:
        ``_weights_fingerprint`` sees those.  Needed after writes through ``p.data`` (a detached alias with its own version
        counter: nothing cheap can see them)."""
        _bump_epoch()
        for m in self.modules():
            if isinstance(m, NativeModule):
                m.__dict__["_weights_version"] = m.__dict__.get("_weights_version", 0) + 1

    def _weights_fingerprint(self):
        """(explicit refresh counter, identity of the tensors, sum of their in-place versions)."""
        st = self.__dict__
        if st.get("_fp_epoch") != _EPOCH[0]:
            ts = [t for t in self.state_dict(keep_vars=True).values() if t.dtype != torch.long]
            st["_fp_tensors"] = ts
            st["_fp_ident"] = hash(tuple((id(t), t.data_ptr()) for t in ts))
            st["_fp_epoch"] = _EPOCH[0]
        v = 0
        for t in st["_fp_tensors"]:
            v += t._version
        return (st["_weights_version"], st["_fp_ident"], v)

    def reuse_output_buffers(self, on=True):
        """Steady-state serving: ``forward`` returns the SAME output tensors on every call of a given geometry (the caller must
        be done with the previous results) instead of allocating ~12 MB per 1280x1280 frame.  Off by default."""
        self.__dict__["_out_cache"] = {} if on else None
        return self

    def parallel_slices(self, n=2):
        """Run a batch as ``n`` equal slices on parallel HIP streams (each slice has its own plan and arena; the slices write into
        the halves of ONE output tensor).  Frames are independent units, so the result is bit-identical to the whole-batch call; the
        point is occupancy: the workgroups of one slice's kernels fill the partly filled last round of tiles of the other's and the
        gaps between dependent launches (skyeye_s bf16 B = 32 @1280: +3.6 % with two slices, measured; four lose again).  Works
        under ``capture_graph`` (fork / join by stream events).  Applies to batch-first modules whose batch divides by n; fp8 plans
        take part only when ``calibrate()`` gave them common calibration frames.  ``n = 1`` switches it off (the default)."""
        if isinstance(n, (list, tuple)):                       # explicit slice sizes, e.g. (20, 12): batches of exactly sum(n) frames are sliced that way
            sizes = tuple(int(v) for v in n)
            if len(sizes) < 1 or any(v < 1 for v in sizes):
                raise ValueError("parallel_slices: slice sizes must be positive")
            self.__dict__["_slice_sizes"] = sizes if len(sizes) > 1 else None
            self.__dict__["_slices"] = len(sizes)
            return self
        self.__dict__["_slice_sizes"] = None
        self.__dict__["_slices"] = max(1, int(n))
        return self

    def set_precision(self, precision):
        """'fp32' (exact MFMA f32 path), 'bf16' (production) or 'fp8' (OCP e4m3 weights and activations with calibrated
        per-tensor scales, bf16 stem; convolutional detector only).  None = follow the parameter dtype like the reference's
        model.half() convention (validate.py:195-197)."""
        if precision not in (None, "fp32", "bf16", "fp8"):
            raise ValueError(precision)
        for m in self.modules():
            if isinstance(m, NativeModule):
                m.__dict__["_precision"] = precision
        return self

    def calibrate(self, *inputs):
        """fp8 engine: the inputs (e.g. a few representative frames, uint8 or float, on the device) from which every plan of
        this module takes its activation scales (one bf16 pass with an amax reduction behind every layer, sky_calibrate).
        Without it an fp8 plan calibrates itself on the first input it sees.  Frames of another size than a plan's are not
        used for it (scales are per tensor, but the statistics of another resolution are another distribution)."""
        frames = [self._prepare_input(t) for t in inputs]
        for m in self.modules():
            if isinstance(m, NativeModule):
                m.__dict__["_calib_inputs"] = None
        self.__dict__["_calib_inputs"] = frames
        for key, (h, fp) in list(self._engines.items()):
            if key[0] == "fp8":
                self._drop_engine(key)
        return self

    def _drop_engine(self, key):
        """Forget a plan and everything that was cached for it (its reusable output tensors).  The handle is closed when its last owner
        lets go of it (``Handle.__del__``): at once here, unless a captured hipGraph still replays into its arena (``graph._sky_keep``)."""
        self._engines.pop(key, None)
        if self._out_cache is not None:
            for k in [k for k in self._out_cache if k == key or (isinstance(k, tuple) and len(k) == 3 and k[0] in ("slot", "sliced") and k[-1] == key)]:
                self._out_cache.pop(k, None)

    def half(self):
        """Reference callers use model.half() for the reduced-precision path (validate.py:195-197, detect.py:107-108).
        Here that selects the bf16 MFMA engine; the fp32 master weights are kept (bf16 has fp32's exponent range)."""
        return self.set_precision("bf16")

    def bfloat16(self):
        return self.set_precision("bf16")

    def float(self):
        self.set_precision("fp32")
        return super().float()

    def _resolved_precision(self):
        if self._precision is not None:
            return self._precision
        for p in self.parameters():
            return "bf16" if p.dtype in (torch.float16, torch.bfloat16) else "fp32"
        return "fp32"

    def _named_weights(self):
        out = {}
        for k, v in self.state_dict().items():
            if v.dtype == torch.long:
                continue
            out[k] = np.ascontiguousarray(v.detach().to("cpu", torch.float32).numpy())
        return out

    def expected_state(self):
        """[(name, shape)] the native engine expects -- must equal the module's own state_dict (tests check)."""
        h = N.Handle(N.make_config(self._sky_module, **self._sky_config()))
        try:
            return h.param_spec()
        finally:
            h.close()

    def _engine(self, inputs, extra_cfg=None):
        return self._engine_entry(inputs, extra_cfg)[1]

    def _engine_entry(self, inputs, extra_cfg=None, slot=0):
        """(cache key, Handle) of the plan for these inputs.  ``slot``: plans of equal geometry that must coexist (batch slices
        running concurrently, ``parallel_slices``)."""
        prec = self._resolved_precision()
        dev = inputs[0].device
        key = (prec, dev.index or 0, tuple(tuple(t.shape) for t in inputs), tuple(sorted((extra_cfg or {}).items())),
               tuple(os.environ.get(k) for k in _PLAN_SWITCHES), slot)
        fp = self._weights_fingerprint()
        pinned = self.__dict__.get("_pinned")                 # keys fetched by the call in progress (_run_sliced): never evicted by it
        if pinned is not None:
            pinned.add(key)
        ent = self._engines.get(key)
        if ent is not None and ent[1] == fp:
            self._engines[key] = self._engines.pop(key)       # least recently USED first (dicts keep insertion order)
            self._keep_for_capture(ent[0])
            return key, ent[0]
        if ent is not None:
            self._drop_engine(key)
        if len(self._engines) >= 6 + self.__dict__.get("_slices", 1):     # keep a few geometries resident (test-time augmentation plans three)
            victim = next((k for k in self._engines if pinned is None or k not in pinned), None)
            if victim is not None:
                self._drop_engine(victim)                     # the least recently used plan goes first
        cfg = dict(self._sky_config())
        cfg.update(extra_cfg or {})
        h = N.Handle(N.make_config(self._sky_module, dtype=N.DTYPES[prec], device=dev.index or 0, **cfg))
        w = self._named_weights()
        if w:
            h.load_weights(w)
        h.plan([N.buffer_from_tensor(t) for t in inputs])
        if prec == "fp8":
            cal = self.__dict__.get("_calib_inputs")
            if not (cal and len(cal) == len(inputs) and all(c.shape[1:] == t.shape[1:] and c.device == t.device for c, t in zip(cal, inputs))):
                cal = inputs                       # self-calibration on the first batch of this geometry
            h.calibrate([N.buffer_from_tensor(t) for t in cal], torch.cuda.current_stream(dev).cuda_stream)
        self._engines[key] = (h, fp)
        self._keep_for_capture(h)
        return key, h

    @staticmethod
    def _keep_for_capture(h):
        """A hipGraph being captured replays into this plan's arena: the graph keeps the handle (``capture_graph``: ``graph._sky_keep``),
        so eviction, ``calibrate()`` or a weight change only drop the cache's reference and the arena lives as long as the graph."""
        from ...utils import metrics as _metrics
        if _metrics._KEEP:
            _metrics._KEEP[-1].append(h)

    def export_engine(self, path, *example_inputs):
        """Write the engine's own weight file for the geometry of ``example_inputs`` (export.py counterpart):
        an .npz with, per packed convolution i, ``conv{i}.weight`` [rows, kpad] (float32, or uint16 = bf16 bits),
        ``conv{i}.bias`` [rows] float32, ``conv{i}.scale`` [rows] float32 and a JSON ``index`` (state-dict name, cout, kernel_size,
        cin, dtype).  BatchNorm is folded (blocks.py:39-41 ``fused_forward``), K is (ky, kx, cin); the stored values are what the
        kernels read: real weight = stored * scale (fp8 row scales; ln 2 for the SiLU layers of the bf16 engine, whose weights and
        bias are kept times log2 e -- include/skyeye_hip.h: sky_packed_scales).  Returns the index."""
        import json
        import numpy as np
        h = self._engine([self._prepare_input(t) for t in example_inputs])
        packed = h.packed_weights()
        arrays, index = {}, []
        for i, p in enumerate(packed):
            arrays[f"conv{i}.weight"], arrays[f"conv{i}.bias"], arrays[f"conv{i}.scale"] = p["weight"], p["bias"], p["scale"]
            index.append(dict(i=i, name=p["name"], cout=p["cout"], kernel_size=p["kernel_size"], cin=p["cin"],
                              dtype="float32" if p["weight"].dtype == np.float32 else "bfloat16"))
        arrays["index"] = np.frombuffer(json.dumps(index).encode(), dtype=np.uint8)
        np.savez(path, **arrays)
        return index

    def _prepare_input(self, t):
        if not torch.is_tensor(t):
            raise TypeError("SkyEye modules take torch tensors")
        if not t.is_cuda:
            raise N.SkyEyeNativeError("SkyEye HIP engine: input tensor is on the CPU; move it to the MI355X "
                                      "(x.to('cuda')) -- the engine has no CPU path")
        if t.dtype in (torch.float16, torch.bfloat16):    # callers that did img.half() (validate.py:237)
            t = t.float()
        return t.contiguous()

    def _run(self, inputs, extra_cfg=None, skip=()):
        """``skip``: indices of optional outputs the caller does not want (the raw detection levels): the engine gets a NULL
        buffer for them and does not write them; the returned list holds None there."""
        inputs = [self._prepare_input(t) for t in inputs]
        nsl = self.__dict__.get("_slices", 1)
        if nsl > 1 and self._sliceable(inputs, nsl):
            return self._run_sliced(inputs, extra_cfg, skip, nsl)
        key, h = self._engine_entry(inputs, extra_cfg)
        cache = self._out_cache
        slot = self.__dict__.get("_out_slot", 0)             # second set of reusable outputs (detect_nms_pipelined: two batches in flight)
        ckey = ("slot", slot, key) if slot else key
        outs = cache.get(ckey) if cache is not None else None
        if outs is None or any((o is None) != (i in skip) for i, o in enumerate(outs)):
            outs = [None if i in skip else torch.empty(s, dtype=torch.float32, device=inputs[0].device)
                    for i, s in enumerate(h.output_shapes())]
            if cache is not None:
                cache[ckey] = outs
        stream = torch.cuda.current_stream(inputs[0].device).cuda_stream
        h.forward([N.buffer_from_tensor(t) for t in inputs],
                  [N.null_buffer() if t is None else N.buffer_from_tensor(t) for t in outs], stream)
        return list(outs)

    def _sliceable(self, inputs, nsl):
        sizes = self.__dict__.get("_slice_sizes")
        if any(t.dim() < 2 or t.shape[0] != inputs[0].shape[0] for t in inputs):
            return False
        if sizes is not None:
            if sum(sizes) != inputs[0].shape[0]:
                return False
        elif inputs[0].shape[0] % nsl or inputs[0].shape[0] < 2 * nsl:
            return False
        if self._resolved_precision() == "fp8":
            # the predicate of _engine_entry: a slice plan takes the common calibration frames only when they have the slices' geometry and
            # device; otherwise every slice would calibrate itself on its own frames (other scales than the whole batch, results that
            # depend on the batch's composition) -> the whole-batch call
            cal = self.__dict__.get("_calib_inputs")
            if not (cal and len(cal) == len(inputs) and all(c.shape[1:] == t.shape[1:] and c.device == t.device for c, t in zip(cal, inputs))):
                return False
        return True

    def _run_sliced(self, inputs, extra_cfg, skip, nsl, post=None):
        """``parallel_slices``: slice i of the batch through plan slot i on side stream i, into rows [i b, (i + 1) b) of the outputs.
        ``post(i, lo, hi, out_views)``: more work of slice i (its NMS) enqueued on its stream before the join."""
        dev = inputs[0].device
        B = inputs[0].shape[0]
        sizes = self.__dict__.get("_slice_sizes") or (B // nsl,) * nsl
        lo = [sum(sizes[:i]) for i in range(nsl)]
        hi = [lo[i] + sizes[i] for i in range(nsl)]
        st = self.__dict__.setdefault("_slice_streams", {})
        streams = st.get(dev.index or 0)
        if streams is None or len(streams) != nsl:
            streams = st[dev.index or 0] = [torch.cuda.Stream(device=dev) for _ in range(nsl)]
        parts = [[t[lo[i]:hi[i]] for t in inputs] for i in range(nsl)]
        self.__dict__["_pinned"] = set()
        try:
            ents = [self._engine_entry(parts[i], extra_cfg, slot=i + 1) for i in range(nsl)]   # (plans are made on the caller's stream)
        finally:
            self.__dict__["_pinned"] = None
        shapes = [(B,) + tuple(sh[1:]) for sh in ents[0][1].output_shapes()]
        cache = self._out_cache
        ckey = ("sliced", (self.__dict__.get("_out_slot", 0), sizes), ents[0][0])
        outs = cache.get(ckey) if cache is not None else None
        if outs is None or any((o is None) != (i in skip) for i, o in enumerate(outs)):
            outs = [None if i in skip else torch.empty(sh, dtype=torch.float32, device=dev) for i, sh in enumerate(shapes)]
            if cache is not None:
                cache[ckey] = outs
        cur = torch.cuda.current_stream(dev)
        for i in range(nsl):
            streams[i].wait_stream(cur)                                         # fork (capturable)
            with torch.cuda.stream(streams[i]):
                ents[i][1].forward([N.buffer_from_tensor(t) for t in parts[i]],
                                   [N.null_buffer() if o is None else N.buffer_from_tensor(o[lo[i]:hi[i]]) for o in outs],
                                   streams[i].cuda_stream)
                if post is not None:
                    post(i, lo[i], hi[i], [None if o is None else o[lo[i]:hi[i]] for o in outs])
        for s_ in streams:
            cur.wait_stream(s_)                                                 # join: the caller's stream owns the results again
        return list(outs)

    def forward(self, x):
        return self._run([x])[0]
