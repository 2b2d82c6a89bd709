"""ctypes binding of libskyeye_hip.so (C ABI declared in include/skyeye_hip.h).

This is the only place the Python package touches native code.  There is NO fallback: if the library is
missing or no HIP device is visible, every compute entry point raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SKYEYE_HIP_LIB: developer switch to load another build of the same library (A/B timing of a kernel change in one session)
LIB_PATH = os.environ.get("SKYEYE_HIP_LIB") or os.path.join(_HERE, "_lib", "libskyeye_hip.so")

SKY_MAX_LEVELS, SKY_MAX_ANCHORS, SKY_MAX_IO = 4, 8, 4
SKY_F32, SKY_BF16, SKY_FP8 = 0, 1, 2
DTYPES = {"fp32": SKY_F32, "bf16": SKY_BF16, "fp8": SKY_FP8}
SKY_IO_F32, SKY_IO_U8 = 0, 1
SKY_NCHW, SKY_NHWC = 0, 1

MODULES = dict(
    DETECTOR=0, ENHANCED_DETECTOR=1, CONV_BLOCK=2, BOTTLENECK=3, CSP=4, SPP=5, FOCUS=6, CHANNEL_ATTENTION=7,
    SPATIAL_ATTENTION=8, COMBINED_ATTENTION=9, BACKBONE=10, NECK=11, HEAD=12, CROSS_LAYER_ATTENTION=13,
    TRANSFORMER_LAYER=14, WINDOWED_ATTENTION=15, DECODE=16, UTILITY=100,
)


class SkyConfig(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("module", ctypes.c_int32), ("dtype", ctypes.c_int32), ("device", ctypes.c_int32),
        ("base_channels", ctypes.c_int32), ("depth_multiple", ctypes.c_float), ("width_multiple", ctypes.c_float),
        ("nc", ctypes.c_int32), ("in_channels", ctypes.c_int32), ("num_levels", ctypes.c_int32), ("num_anchors", ctypes.c_int32),
        ("anchors", ctypes.c_float * (SKY_MAX_LEVELS * SKY_MAX_ANCHORS * 2)),
        ("c_in", ctypes.c_int32), ("c_out", ctypes.c_int32), ("kernel_size", ctypes.c_int32), ("stride", ctypes.c_int32),
        ("activation", ctypes.c_int32), ("num_blocks", ctypes.c_int32), ("shortcut", ctypes.c_int32), ("expansion", ctypes.c_float),
        ("heads", ctypes.c_int32), ("window_size", ctypes.c_int32), ("region_size", ctypes.c_int32),
        ("reduction_ratio", ctypes.c_int32), ("key_channels", ctypes.c_int32),
        ("level_channels", ctypes.c_int32 * SKY_MAX_LEVELS), ("input_h", ctypes.c_int32), ("input_w", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 6),
    ]


class SkyTensorDesc(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.c_void_p), ("ndim", ctypes.c_int32), ("shape", ctypes.c_int64 * 4)]


class SkyBuffer(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("dtype", ctypes.c_int32), ("layout", ctypes.c_int32), ("ndim", ctypes.c_int32),
                ("shape", ctypes.c_int64 * 5)]


class SkyNmsParams(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("conf_threshold", ctypes.c_float), ("iou_threshold", ctypes.c_float),
                ("agnostic", ctypes.c_int32), ("multi_label", ctypes.c_int32), ("max_detections", ctypes.c_int32),
                ("max_nms", ctypes.c_int32), ("max_wh", ctypes.c_float), ("mode", ctypes.c_int32), ("n_classes", ctypes.c_int32),
                ("classes", ctypes.c_int32 * 64), ("out_image_stride", ctypes.c_int32), ("counts_stride", ctypes.c_int32)]


# every symbol include/skyeye_hip.h declares (tests/test_abi.py checks the library exports all of them)
SYMBOLS = ["sky_abi_version", "sky_build_info", "sky_device_count", "sky_last_error", "sky_create", "sky_destroy", "sky_num_params",
           "sky_param_info", "sky_load_weights", "sky_plan", "sky_num_outputs", "sky_output_info", "sky_forward", "sky_nms",
           "sky_nms_fetch", "sky_box_iou", "sky_letterbox", "sky_scale_img", "sky_map_detections", "sky_offset_boxes", "sky_tile_gather", "sky_num_packed", "sky_packed_info", "sky_packed_read", "sky_packed_scales", "sky_calibrate", "sky_num_scales", "sky_scales_read", "sky_scales_write", "sky_plan_stats", "sky_time_forward", "sky_profile_forward", "sky_op_info", "sky_op_bytes", "sky_op_io_bytes"]

_lib = None


class SkyPackedDesc(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 128), ("rows", ctypes.c_int32), ("cout", ctypes.c_int32), ("kpad", ctypes.c_int32),
                ("kernel_size", ctypes.c_int32), ("cin", ctypes.c_int32), ("dtype", ctypes.c_int32)]


class SkyEyeNativeError(RuntimeError):
    pass


def lib():
    """Load libskyeye_hip.so or fail loudly (no CPU path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SkyEyeNativeError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` (hipcc --offload-arch=gfx950). "
            "The SkyEye engine has no CPU or PyTorch fallback.")
    try:
        # the process's HIP runtime must be the one torch brings (its bundled libamdhip64): loaded first, this library would pull in /opt/rocm's copy,
        # torch's copy would initialise a second runtime and the devices / streams / pointers of the two would not be each other's
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, ip = ctypes.c_void_p, ctypes.c_int
    L.sky_abi_version.restype = ip
    L.sky_build_info.restype = ctypes.c_char_p
    L.sky_device_count.restype = ip
    L.sky_last_error.restype = ctypes.c_char_p
    L.sky_last_error.argtypes = [vp]
    L.sky_create.argtypes = [ctypes.POINTER(SkyConfig), ctypes.POINTER(vp)]
    L.sky_destroy.argtypes = [vp]
    L.sky_destroy.restype = None
    L.sky_num_params.argtypes = [vp]
    L.sky_param_info.argtypes = [vp, ip, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)]
    L.sky_load_weights.argtypes = [vp, ctypes.POINTER(SkyTensorDesc), ip]
    L.sky_plan.argtypes = [vp, ip, ctypes.POINTER(SkyBuffer)]
    L.sky_num_outputs.argtypes = [vp]
    L.sky_output_info.argtypes = [vp, ip, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)]
    L.sky_forward.argtypes = [vp, ip, ctypes.POINTER(SkyBuffer), ip, ctypes.POINTER(SkyBuffer), vp]
    L.sky_time_forward.argtypes = [vp, ip, ctypes.POINTER(SkyBuffer), ip, ctypes.POINTER(SkyBuffer), vp, ip, ctypes.POINTER(ctypes.c_float)]
    L.sky_plan_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                 ctypes.POINTER(ctypes.c_int32)]
    L.sky_profile_forward.argtypes = [vp, ip, ctypes.POINTER(SkyBuffer), ip, ctypes.POINTER(SkyBuffer), vp, ip, ip,
                                      ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32),
                                      ctypes.POINTER(ctypes.c_int32)]
    L.sky_op_info.argtypes = [vp, ip, ctypes.c_char_p, ip]
    L.sky_op_bytes.argtypes = [vp, ip, ctypes.POINTER(ctypes.c_double)]
    L.sky_op_io_bytes.argtypes = [vp, ip, ip, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.sky_nms.argtypes = [vp, vp, ip, ip, ip, ctypes.POINTER(SkyNmsParams), vp, vp, vp]
    L.sky_nms_fetch.argtypes = [vp, vp, ip, vp, vp]
    L.sky_box_iou.argtypes = [vp, vp, ip, ip, vp, ip, vp, vp]
    L.sky_num_packed.argtypes = [vp]
    L.sky_packed_info.argtypes = [vp, ip, ctypes.POINTER(SkyPackedDesc)]
    L.sky_packed_read.argtypes = [vp, ip, vp, ctypes.c_size_t, vp, ctypes.c_size_t]
    L.sky_packed_scales.argtypes = [vp, ip, vp, ctypes.c_size_t]
    L.sky_calibrate.argtypes = [vp, ip, ctypes.POINTER(SkyBuffer), vp]
    L.sky_num_scales.argtypes = [vp]
    L.sky_scales_read.argtypes = [vp, vp, ip]
    L.sky_scales_write.argtypes = [vp, vp, ip]
    L.sky_letterbox.argtypes = [vp, vp, ip, ip, vp, ip, ip, ip, ip, ip, ip, ip, ip, ip, vp]
    fp = ctypes.c_float
    L.sky_scale_img.argtypes = [vp, vp, ip, ip, ip, ip, ip, vp, ip, ip, ip, ip, ip, fp, vp]
    L.sky_map_detections.argtypes = [vp, vp, ip, ip, ip, ip, ip, fp, ip, fp, fp, vp, ip, vp, ctypes.c_int64, ctypes.c_int64, vp]
    L.sky_offset_boxes.argtypes = [vp, vp, vp, ip, ip, ip, vp, vp]
    L.sky_tile_gather.argtypes = [vp, vp, ip, ip, ip, vp, ip, vp, ip, ip, ip, ip, vp]
    _lib = L
    return L


def build_info():
    """Hash of the sources the LOADED library was built from (sky_build_info)."""
    return lib().sky_build_info().decode()


def source_hash():
    """The same hash computed from the working tree (csrc/Makefile: build_hash.h), or None where csrc/ did not travel."""
    import glob
    import hashlib
    csrc = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
    if not os.path.exists(os.path.join(csrc, "engine.cpp")):
        return None
    names = [os.path.basename(f) for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))]
    names = sorted(set(n for n in names if n != "build_hash.h") | {"engine.cpp", "Makefile", "../../include/skyeye_hip.h"})
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(csrc, n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def check(rc, handle=None):
    if rc != 0:
        msg = lib().sky_last_error(handle)
        raise SkyEyeNativeError(f"libskyeye_hip: {msg.decode() if msg else 'error'} (status {rc})")


def make_config(module, dtype=SKY_F32, device=0, **kw):
    cfg = SkyConfig()
    cfg.struct_size = ctypes.sizeof(SkyConfig)
    cfg.module = MODULES[module]
    cfg.dtype = dtype
    cfg.device = device
    anchors = kw.pop("anchors", None)
    level_channels = kw.pop("level_channels", None)
    cfg.reserved[0] = int(bool(kw.pop("head_attention", False)))     # D5 wiring (include/skyeye_hip.h, sky_config.reserved)
    for k, v in kw.items():
        setattr(cfg, k, v)
    if anchors is not None:
        cfg.num_levels = len(anchors)
        cfg.num_anchors = len(anchors[0])
        flat = [float(v) for lvl in anchors for a in lvl for v in a]
        for i, v in enumerate(flat):
            cfg.anchors[i] = v
    if level_channels is not None:
        for i, v in enumerate(level_channels):
            cfg.level_channels[i] = int(v)
    return cfg


class Handle:
    """Owns one sky_handle*."""

    def __init__(self, cfg):
        self.L = lib()
        self.h = ctypes.c_void_p()
        rc = self.L.sky_create(ctypes.byref(cfg), ctypes.byref(self.h))
        if rc != 0:
            msg = self.L.sky_last_error(None)
            raise SkyEyeNativeError(f"sky_create: {msg.decode() if msg else 'error'} (status {rc})")

    def close(self):
        if getattr(self, "h", None):
            self.L.sky_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def param_spec(self):
        out = []
        name, nd, shp = ctypes.c_char_p(), ctypes.c_int32(), (ctypes.c_int64 * 4)()
        for i in range(self.L.sky_num_params(self.h)):
            check(self.L.sky_param_info(self.h, i, ctypes.byref(name), ctypes.byref(nd), shp), self.h)
            out.append((name.value.decode(), tuple(int(shp[k]) for k in range(nd.value))))
        return out

    def load_weights(self, named_arrays):
        """named_arrays: {name: C-contiguous float32 numpy array}"""
        keep, descs = [], (SkyTensorDesc * len(named_arrays))()
        for i, (k, a) in enumerate(named_arrays.items()):
            kb = k.encode()
            keep.append((kb, a))
            descs[i].name = kb
            descs[i].data = a.ctypes.data
            descs[i].ndim = a.ndim
            for d in range(a.ndim):
                descs[i].shape[d] = a.shape[d]
        check(self.L.sky_load_weights(self.h, descs, len(named_arrays)), self.h)

    def plan(self, bufs):
        arr = (SkyBuffer * len(bufs))(*bufs)
        check(self.L.sky_plan(self.h, len(bufs), arr), self.h)

    def output_shapes(self):
        out = []
        nd, shp = ctypes.c_int32(), (ctypes.c_int64 * 5)()
        for i in range(self.L.sky_num_outputs(self.h)):
            check(self.L.sky_output_info(self.h, i, ctypes.byref(nd), shp), self.h)
            out.append(tuple(int(shp[k]) for k in range(nd.value)))
        return out

    def forward(self, ins, outs, stream):
        a = (SkyBuffer * len(ins))(*ins)
        b = (SkyBuffer * max(len(outs), 1))(*outs)
        check(self.L.sky_forward(self.h, len(ins), a, len(outs), b, ctypes.c_void_p(stream)), self.h)

    def calibrate(self, ins, stream):
        """fp8 engine: activation scales from one bf16 pass over ``ins`` (sky_calibrate)."""
        a = (SkyBuffer * len(ins))(*ins)
        check(self.L.sky_calibrate(self.h, len(ins), a, ctypes.c_void_p(stream)), self.h)

    def scales(self):
        """-> float32 array, one scale per workspace buffer of the plan (1 for non-fp8 buffers)."""
        import numpy as np
        n = self.L.sky_num_scales(self.h)
        out = np.ones((n,), np.float32)
        if n:
            check(self.L.sky_scales_read(self.h, out.ctypes.data_as(ctypes.c_void_p), n), self.h)
        return out

    def set_scales(self, scales):
        import numpy as np
        a = np.ascontiguousarray(scales, np.float32)
        check(self.L.sky_scales_write(self.h, a.ctypes.data_as(ctypes.c_void_p), a.size), self.h)

    def time_forward(self, ins, outs, stream, iters):
        a = (SkyBuffer * len(ins))(*ins)
        b = (SkyBuffer * max(len(outs), 1))(*outs)
        ms = ctypes.c_float()
        check(self.L.sky_time_forward(self.h, len(ins), a, len(outs), b, ctypes.c_void_p(stream), iters, ctypes.byref(ms)), self.h)
        return ms.value

    def packed_weights(self):
        """-> list of dict(name, cout, kernel_size, cin, weight [rows, kpad] (float32 | uint16 bf16 bits | uint8 e4m3 bytes), bias [rows]
        float32, scale [rows] float32 (per-output-channel weight scale of fp8 weights, ones otherwise), dtype):
        the BatchNorm-folded, layout-converted convolution weights exactly as the kernels read them (sky_packed_*)."""
        import numpy as np
        out = []
        n = self.L.sky_num_packed(self.h)
        if n < 0:
            check(n, self.h)
        for i in range(n):
            d = SkyPackedDesc()
            check(self.L.sky_packed_info(self.h, i, ctypes.byref(d)), self.h)
            w = np.empty((d.rows, d.kpad), {SKY_F32: np.float32, SKY_BF16: np.uint16, SKY_FP8: np.uint8}[d.dtype])
            b = np.empty((d.rows,), np.float32)
            sc = np.empty((d.rows,), np.float32)
            check(self.L.sky_packed_read(self.h, i, w.ctypes.data_as(ctypes.c_void_p), w.nbytes, b.ctypes.data_as(ctypes.c_void_p), b.size), self.h)
            check(self.L.sky_packed_scales(self.h, i, sc.ctypes.data_as(ctypes.c_void_p), sc.size), self.h)
            out.append(dict(name=d.name.decode(), cout=d.cout, kernel_size=d.kernel_size, cin=d.cin, weight=w, bias=b, scale=sc, dtype=d.dtype))
        return out

    def profile_forward(self, ins, outs, stream, iters=3, max_ops=4096):
        """-> list of (ms, flops, tag) per launch of the planned graph."""
        a = (SkyBuffer * len(ins))(*ins)
        b = (SkyBuffer * max(len(outs), 1))(*outs)
        ms, fl, tg, n = (ctypes.c_float * max_ops)(), (ctypes.c_double * max_ops)(), (ctypes.c_int32 * max_ops)(), ctypes.c_int32()
        check(self.L.sky_profile_forward(self.h, len(ins), a, len(outs), b, ctypes.c_void_p(stream), iters, max_ops, ms, fl, tg,
                                         ctypes.byref(n)), self.h)
        return [(ms[i], fl[i], tg[i]) for i in range(n.value)]

    def op_info(self, i):
        buf = ctypes.create_string_buffer(256)
        check(self.L.sky_op_info(self.h, i, buf, 256), self.h)
        return buf.value.decode()

    def op_bytes(self, i):
        b = ctypes.c_double()
        check(self.L.sky_op_bytes(self.h, i, ctypes.byref(b)), self.h)
        return b.value

    def op_io_bytes(self, i, with_raw=False):
        """(bytes read, bytes written) of planned op i; a launch that computes a chain of ops reads the first op's and writes the last op's"""
        r, w = ctypes.c_double(), ctypes.c_double()
        check(self.L.sky_op_io_bytes(self.h, i, 1 if with_raw else 0, ctypes.byref(r), ctypes.byref(w)), self.h)
        return r.value, w.value

    def stats(self):
        f, a, w, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int32()
        check(self.L.sky_plan_stats(self.h, ctypes.byref(f), ctypes.byref(a), ctypes.byref(w), ctypes.byref(n)), self.h)
        return dict(flops=f.value, activation_bytes=a.value, weight_bytes=w.value, launches=n.value)


def null_buffer():
    """An optional output the caller does not want (sky_forward: raw detection levels may be NULL)."""
    return SkyBuffer()


def buffer_from_tensor(t, layout=SKY_NCHW):
    """Describe a torch CUDA tensor (float32 or uint8, contiguous) as a sky_buffer."""
    import torch
    if not t.is_cuda:
        raise SkyEyeNativeError("SkyEye HIP engine: tensors must live on a HIP device (there is no CPU path)")
    if not t.is_contiguous():
        raise SkyEyeNativeError("SkyEye HIP engine: tensors must be contiguous")
    b = SkyBuffer()
    b.data = t.data_ptr()
    if t.dtype == torch.float32:
        b.dtype = SKY_IO_F32
    elif t.dtype == torch.uint8:
        b.dtype = SKY_IO_U8
    else:
        raise SkyEyeNativeError(f"unsupported boundary dtype {t.dtype}")
    b.layout = layout
    b.ndim = t.dim()
    for i, s in enumerate(t.shape):
        b.shape[i] = s
    return b
