"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement of the reference's detection forward pass, one function per
reference module, in the reference's layout (NCHW fp32 numpy arrays).  Heavy
operators run in ``libsky_oracle.so`` (plain C + OpenMP, built from
``sky_oracle_ops.c`` / ``sky_oracle_nms.c`` by ``oracle/Makefile``); everything
else is numpy.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product
(``skyeye`` package + ``libskyeye_hip.so``) never does.

Parity status: PINNED for every function below by ``tests/golden/*.npz``,
which ``tests/golden/make_golden.py`` produced by executing the reference's
own PyTorch classes in the build container (``tests/test_oracle_golden.py``),
EXCEPT the greedy suppression inside ``non_max_suppression`` (third-party
``torchvision.ops.nms``, absent and un-pinned upstream): PARITY UNPINNED there,
see ``sky_oracle_nms.c``.  Front-end helpers at the end of the file: ``scale_img`` is pinned by
``tests/golden/tta.npz`` (the reference's own function, 5e-7 absolute); ``letterbox_pixels`` restates OpenCV's
fixed-point resize and is UNPINNED (cv2 is not installed, the reference holds no image fixtures); ``map_detections`` /
``tta_forward`` / ``tile_origins`` / ``tile_gather`` are build-defined (the reference has the ``augment=`` flag but no
body for it, and no tiling) and serve as the arithmetic the HIP kernels are compared with bit for bit.

Weights arrive as ``{state_dict_name: ndarray}`` with the reference's names
 and a prefix per sub-module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_F = ctypes.POINTER(ctypes.c_float)


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libsky_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(so)
        _LIB.sky_oracle_threads.restype = ctypes.c_int
        _LIB.sky_oracle_nms_image.restype = ctypes.c_int
    return _LIB


def threads():
    return int(_lib().sky_oracle_threads())


def set_threads(n):
    _lib().sky_oracle_set_threads(ctypes.c_int(int(n)))


def _p(a):
    return a.ctypes.data_as(_F)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------------------- operators
def conv2d(x, w, bias=None, stride=1, pad=None):
    """nn.Conv2d forward (blocks.py:31, detector.py:56-59, attention.py:79,167-170)."""
    x, w = _c(x), _c(w)
    B, Cin, H, W = x.shape
    Cout, Cin2, K, K2 = w.shape
    assert Cin == Cin2 and K == K2
    pad = K // 2 if pad is None else pad
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    y = np.empty((B, Cout, Ho, Wo), np.float32)
    b = None if bias is None else _c(bias)
    _lib().sky_oracle_conv2d(_p(x), _p(w), None if b is None else _p(b), _p(y), B, Cin, H, W, Cout, K, stride, pad)
    return y


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float32)))).astype(np.float32)


def maxpool2d(x, k):
    """nn.MaxPool2d(k, stride=1, padding=k//2) (blocks.py:142-144)."""
    x = _c(x)
    y = np.empty_like(x)
    B, C, H, W = x.shape
    _lib().sky_oracle_maxpool2d(_p(x), _p(y), B, C, H, W, k)
    return y


def upsample_nearest(x, size):
    """F.interpolate(size=..., mode='nearest') (detector.py:214,218)."""
    x = _c(x)
    B, C, H, W = x.shape
    y = np.empty((B, C, size[0], size[1]), np.float32)
    _lib().sky_oracle_upsample_nearest(_p(x), _p(y), B, C, H, W, size[0], size[1])
    return y


def bilinear(x, size):
    """F.interpolate(size=..., mode='bilinear', align_corners=False) (attention.py:211-212)."""
    x = _c(x)
    B, C, H, W = x.shape
    y = np.empty((B, C, size[0], size[1]), np.float32)
    _lib().sky_oracle_bilinear(_p(x), _p(y), B, C, H, W, size[0], size[1])
    return y


def linear(x, w, bias=None):
    """nn.Linear on the last axis."""
    shp = x.shape
    x2 = _c(x.reshape(-1, shp[-1]))
    w = _c(w)
    y = np.empty((x2.shape[0], w.shape[0]), np.float32)
    b = None if bias is None else _c(bias)
    _lib().sky_oracle_linear(_p(x2), _p(w), None if b is None else _p(b), _p(y), x2.shape[0], x2.shape[1], w.shape[0])
    return y.reshape(shp[:-1] + (w.shape[0],))


# --------------------------------------------------------------------------- blocks.py
def conv_block(P, pre, x, k, stride=1, act=True):
    """ConvolutionBlock.forward: act(bn(conv(x)))  (blocks.py:10-37); BN eps = 1e-5 (eval)."""
    y = conv2d(x, P[pre + "conv.weight"], None, stride, k // 2)
    B, C, H, W = y.shape
    _lib().sky_oracle_bn_act(_p(y), _p(_c(P[pre + "bn.weight"])), _p(_c(P[pre + "bn.bias"])),
                             _p(_c(P[pre + "bn.running_mean"])), _p(_c(P[pre + "bn.running_var"])),
                             ctypes.c_float(1e-5), B, C, H * W, 1 if act else 0)
    return y


def _ksize(P, pre):
    return int(P[pre + "conv.weight"].shape[-1])


def bottleneck(P, pre, x, shortcut=True):
    """BottleneckBlock.forward: x + cv2(cv1(x)) iff shortcut and cin == cout (blocks.py:69-90)."""
    y = conv_block(P, pre + "cv2.", conv_block(P, pre + "cv1.", x, 1), 3)
    use = shortcut and P[pre + "cv1.conv.weight"].shape[1] == P[pre + "cv2.conv.weight"].shape[0]
    return x + y if use else y


def csp(P, pre, x):
    """CSPBlock.forward: cv3(cat(bottlenecks(cv1(x)), cv2(x)))  (blocks.py:93-123)."""
    y1 = conv_block(P, pre + "cv1.", x, 1)
    j = 0
    while f"{pre}bottlenecks.{j}.cv1.conv.weight" in P:
        y1 = bottleneck(P, f"{pre}bottlenecks.{j}.", y1)
        j += 1
    y2 = conv_block(P, pre + "cv2.", x, 1)
    return conv_block(P, pre + "cv3.", np.concatenate((y1, y2), 1), 1)


def spp(P, pre, x, kernel_sizes=(5, 9, 13)):
    """SPPBlock.forward: cv2(cat([y] + [maxpool_k(y)]))  (blocks.py:126-149)."""
    y = conv_block(P, pre + "cv1.", x, 1)
    return conv_block(P, pre + "cv2.", np.concatenate([y] + [maxpool2d(y, k) for k in kernel_sizes], 1), 1)


def focus(P, pre, x):
    """FocusBlock.forward: space-to-depth in the order TL, BL, TR, BR then ConvBlock (blocks.py:152-182)."""
    patches = [x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]]
    return conv_block(P, pre + "conv.", np.concatenate(patches, 1), _ksize(P, pre + "conv."))


# --------------------------------------------------------------------------- attention.py
def channel_attention(P, pre, x):
    """ChannelAttention.forward (attention.py:11-60): x * sigmoid(mlp(avg) + mlp(max)), bias-free MLP + ReLU."""
    avg = x.mean(axis=(2, 3), dtype=np.float32)
    mx = x.max(axis=(2, 3))
    w0, w2 = P[pre + "shared_mlp.0.weight"], P[pre + "shared_mlp.2.weight"]

    def mlp(v):
        return linear(np.maximum(linear(v, w0), 0.0), w2)

    att = sigmoid(mlp(avg) + mlp(mx))
    return (x * att[:, :, None, None]).astype(np.float32)


def spatial_attention(P, pre, x):
    """SpatialAttention.forward (attention.py:63-98): x * sigmoid(conv7x7(cat(mean_c, max_c)))."""
    avg = x.mean(axis=1, keepdims=True, dtype=np.float32)
    mx = x.max(axis=1, keepdims=True)
    a = sigmoid(conv2d(np.concatenate([avg, mx], 1), P[pre + "conv.weight"], None, 1, 3))
    return (x * a).astype(np.float32)


def combined_attention(P, pre, x):
    """CombinedAttention.forward (attention.py:101-130): channel then spatial."""
    return spatial_attention(P, pre + "spatial_attention.", channel_attention(P, pre + "channel_attention.", x))


def cross_layer_attention(P, pre, query, key, heads=4, region_size=2):
    """CrossLayerAttention.forward (attention.py:174-241), closed form of SURVEY App. B.9:
    the region_size^2 'patches' are identical bilinear resamples (attention.py:208-215), the softmax
    runs over dim=3 = image rows (attention.py:172,232), scale = 1/sqrt(query_channels) (attention.py:159)."""
    q = conv2d(query, P[pre + "query_projection.weight"], P[pre + "query_projection.bias"], 1, 0)
    k = conv2d(key, P[pre + "key_projection.weight"], P[pre + "key_projection.bias"], 1, 0)
    v = conv2d(key, P[pre + "value_projection.weight"], P[pre + "value_projection.bias"], 1, 0)
    B, Cq, H, W = q.shape
    k = bilinear(k, (H, W))
    v = bilinear(v, (H, W))
    scale = np.float32(1.0 / np.sqrt(np.float32(query.shape[1])))
    d = Cq // heads
    s = (q.reshape(B, heads, d, H, W) * k.reshape(B, heads, d, H, W)).sum(2, dtype=np.float32) * scale   # [B,h,H,W]
    s = s - s.max(axis=2, keepdims=True)
    e = np.exp(s)
    a = e / e.sum(axis=2, keepdims=True, dtype=np.float32)                                             # softmax over H
    R2 = np.float32(region_size * region_size)
    out = (a[:, :, None] * v.reshape(B, heads, d, H, W)) * R2
    out = out.reshape(B, Cq, H, W).astype(np.float32)
    return conv2d(out, P[pre + "output_projection.weight"], P[pre + "output_projection.bias"], 1, 0)


def _layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True, dtype=np.float32)
    var = ((x - mu) ** 2).mean(-1, keepdims=True, dtype=np.float32)
    return ((x - mu) / np.sqrt(var + np.float32(eps)) * w + b).astype(np.float32)


def _softmax(x, axis=-1):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return (e / e.sum(axis=axis, keepdims=True, dtype=np.float32)).astype(np.float32)


def transformer_layer(P, pre, x, num_heads):
    """TransformerLayer.forward in eval mode (attention.py:282-309): pre-LN MHA + FFN(ReLU), tokens = y*W+x."""
    B, C, H, W = x.shape
    t = x.reshape(B, C, H * W).transpose(0, 2, 1)                                                      # [B,N,C]
    n1 = _layer_norm(t, P[pre + "norm1.weight"], P[pre + "norm1.bias"])
    qkv = linear(n1, P[pre + "self_attn.in_proj_weight"], P[pre + "self_attn.in_proj_bias"])
    q, k, v = np.split(qkv, 3, axis=-1)
    d = C // num_heads

    def heads_(a):
        return a.reshape(B, H * W, num_heads, d).transpose(0, 2, 1, 3)

    q, k, v = heads_(q) * np.float32(1.0 / np.sqrt(d)), heads_(k), heads_(v)
    a = _softmax(q @ k.transpose(0, 1, 3, 2))
    o = (a @ v).transpose(0, 2, 1, 3).reshape(B, H * W, C)
    t = t + linear(o, P[pre + "self_attn.out_proj.weight"], P[pre + "self_attn.out_proj.bias"])
    n2 = _layer_norm(t, P[pre + "norm2.weight"], P[pre + "norm2.bias"])
    f = np.maximum(linear(n2, P[pre + "feedforward.0.weight"], P[pre + "feedforward.0.bias"]), 0.0)
    t = t + linear(f, P[pre + "feedforward.3.weight"], P[pre + "feedforward.3.bias"])
    return t.transpose(0, 2, 1).reshape(B, C, H, W).astype(np.float32)


def relative_position_index(ws):
    """attention.py:342-353 (meshgrid default 'ij')."""
    coords = np.stack(np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij"))
    cf = coords.reshape(2, -1)
    rel = (cf[:, :, None] - cf[:, None, :]).transpose(1, 2, 0).copy()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def windowed_self_attention(P, pre, x, window_size, num_heads, mask=None):
    """WindowedSelfAttention.forward (attention.py:358-399); x is already windowed [B*nW, ws*ws, C]."""
    B_, N, C = x.shape
    d = C // num_heads
    qkv = linear(x, P[pre + "qkv.weight"], P[pre + "qkv.bias"]).reshape(B_, N, 3, num_heads, d).transpose(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * np.float32(d ** -0.5), qkv[1], qkv[2]
    attn = q @ k.transpose(0, 1, 3, 2)
    idx = relative_position_index(window_size).reshape(-1)
    bias = P[pre + "relative_position_bias_table"][idx].reshape(N, N, -1).transpose(2, 0, 1)
    attn = attn + bias[None]
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.reshape(B_ // nW, nW, num_heads, N, N) + mask[None, :, None]
        attn = attn.reshape(-1, num_heads, N, N)
    attn = _softmax(attn.astype(np.float32))
    o = (attn @ v).transpose(0, 2, 1, 3).reshape(B_, N, C)
    return linear(o, P[pre + "proj.weight"], P[pre + "proj.bias"])


# --------------------------------------------------------------------------- backbone.py
def backbone(P, pre, x):
    """Backbone.forward (backbone.py:82-99): stage1..4, returns [s2, s3, s4]."""
    s = focus(P, pre + "stage1.0.", x)                                   # backbone.py:48
    s = conv_block(P, pre + "stage1.1.", s, 3, 2)                        # :50
    s1 = csp(P, pre + "stage1.2.", s)                                    # :52
    s2 = csp(P, pre + "stage2.1.", conv_block(P, pre + "stage2.0.", s1, 3, 2))      # :56-61
    s3 = csp(P, pre + "stage3.1.", conv_block(P, pre + "stage3.0.", s2, 3, 2))      # :64-68
    s3 = combined_attention(P, pre + "stage3.2.", s3)                    # :70
    s4 = csp(P, pre + "stage4.1.", conv_block(P, pre + "stage4.0.", s3, 3, 2))      # :74-78
    s4 = spp(P, pre + "stage4.2.", s4)                                   # :79
    return [s2, s3, s4]


# --------------------------------------------------------------------------- detector.py
def feature_neck(P, pre, feats):
    """FeatureNeck.forward (detector.py:197-231), quirks kept: lateral_conv4 reads RAW p4 (:211),
    p5_cat concatenates RAW p5 (:228)."""
    p3, p4, p5 = feats
    p5_td = conv_block(P, pre + "lateral_conv5.", p5, 1)
    p4_td = conv_block(P, pre + "lateral_conv4.", p4, 1)
    p4_processed = csp(P, pre + "fpn_conv4.", np.concatenate([upsample_nearest(p5_td, p4.shape[2:]), p4], 1))
    p3_processed = csp(P, pre + "fpn_conv3.", np.concatenate([upsample_nearest(p4_td, p3.shape[2:]), p3], 1))
    p4_out = csp(P, pre + "pan_conv4.", np.concatenate([conv_block(P, pre + "downsample3.", p3_processed, 3, 2),
                                                       p4_processed], 1))
    p5_out = csp(P, pre + "pan_conv5.", np.concatenate([conv_block(P, pre + "downsample4.", p4_out, 3, 2), p5], 1))
    return [p3_processed, p4_out, p5_out]


DEFAULT_ANCHORS = [[[10, 13], [16, 30], [33, 23]], [[30, 61], [62, 45], [59, 119]],
                   [[116, 90], [156, 198], [373, 326]]]                   # detector.py:39-43


def detection_head(P, pre, feats, num_outputs, num_anchors):
    """DetectionHead.forward (detector.py:61-86): 1x1 conv with bias, view(B,na,no,gh,gw).permute(0,1,3,4,2)."""
    outs = []
    for i, f in enumerate(feats):
        y = conv2d(f, P[f"{pre}detection_layers.{i}.weight"], P[f"{pre}detection_layers.{i}.bias"], 1, 0)
        B, _, gh, gw = y.shape
        outs.append(np.ascontiguousarray(y.reshape(B, num_anchors, num_outputs, gh, gw).transpose(0, 1, 3, 4, 2)))
    return outs


def process_detections(outputs, input_shape, anchors):
    """DetectionHead.process_detections (detector.py:88-145).  Kept literally: stride = max(H/gh, W/gw)
    (:107-109) and anchors (pixels) are multiplied by the stride again (:119-121)."""
    dets = []
    for i, out in enumerate(outputs):
        B, na, gh, gw, no = out.shape
        stride = np.float32(max(input_shape[0] / gh, input_shape[1] / gw))
        yv, xv = np.meshgrid(np.arange(gh), np.arange(gw), indexing="ij")
        grid = np.stack((xv, yv), 2).reshape(1, 1, gh, gw, 2).astype(np.float32)
        anchor_grid = (np.asarray(anchors[i], dtype=np.float32).reshape(1, na, 1, 1, 2) * stride).astype(np.float32)
        y = sigmoid(out)
        y[..., 0:2] = (y[..., 0:2] * np.float32(2) - np.float32(0.5) + grid) * stride
        y[..., 2:4] = (y[..., 2:4] * np.float32(2)) ** 2 * anchor_grid
        dets.append(y.reshape(B, -1, no))
    return np.concatenate(dets, 1)


def windowed_map(P, pre, x, window_size, num_heads):
    """window_partition -> WindowedSelfAttention -> window_reverse on a [B, C, H, W] map (D5 wiring)."""
    B, C, H, W = x.shape
    ws = window_size
    t = x.transpose(0, 2, 3, 1).reshape(B, H // ws, ws, W // ws, ws, C).transpose(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    t = windowed_self_attention(P, pre, np.ascontiguousarray(t), ws, num_heads)
    t = t.reshape(B, H // ws, W // ws, ws, ws, C).transpose(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
    return np.ascontiguousarray(t.transpose(0, 3, 1, 2)).astype(np.float32)


def detector_forward(P, x, nc, anchors=None, enhanced=False, head_attention=False):
    """SkyEyeDetector.forward in eval mode (detector.py:300-324) composed per SURVEY App. A D1/D2;
    EnhancedSkyEyeDetector.forward (detector.py:471-501) with D4 when ``enhanced``; ``head_attention``: D5 wiring
    (WindowedSelfAttention(8, C/32 heads) on P3 / P4, TransformerLayer(8 heads) on P5 ahead of the detection convs)."""
    anchors = DEFAULT_ANCHORS if anchors is None else anchors
    feats = backbone(P, "backbone.backbone.", x)
    neck = feature_neck(P, "neck.", feats)
    if enhanced:
        p3, p4, p5 = neck
        p4e = cross_layer_attention(P, "cross_attention_p5_p4.", p4, p5) + p4
        p3e = cross_layer_attention(P, "cross_attention_p4_p3.", p3, p4e) + p3
        neck = [p3e, p4e, p5]
    if head_attention:
        p3, p4, p5 = neck
        neck = [windowed_map(P, "head_attention.p3.", p3, 8, p3.shape[1] // 32), windowed_map(P, "head_attention.p4.", p4, 8, p4.shape[1] // 32),
                transformer_layer(P, "head_attention.p5.", p5, 8)]
    raw = detection_head(P, "detection_head.", neck, nc + 5, len(anchors[0]))
    det = process_detections([r.copy() for r in raw], x.shape[2:], anchors)
    return det, raw


# --------------------------------------------------------------------------- metrics.py
def non_max_suppression(prediction, conf_threshold=0.25, iou_threshold=0.45, classes=None, agnostic=False,
                        multi_label=False, max_detections=300, mode="literal"):
    """non_max_suppression (metrics.py:361-457) -> list of [n, 6|7] arrays, one per image."""
    pred = _c(prediction)
    B, N, no = pred.shape
    nc = no - 5
    cls = None if classes is None else np.asarray(classes, dtype=np.int32)
    out = []
    buf = np.empty((max_detections, 7), np.float32)
    cols = ctypes.c_int(0)
    for b in range(B):
        n = _lib().sky_oracle_nms_image(
            _p(pred[b]), N, nc, ctypes.c_float(conf_threshold), ctypes.c_float(iou_threshold),
            None if cls is None else cls.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), 0 if cls is None else len(cls),
            int(bool(agnostic)), int(bool(multi_label)), int(max_detections), 30000, ctypes.c_float(4096.0),
            {"literal": 0, "corrected": 1, "rows": 2}[mode], _p(buf), ctypes.byref(cols))
        c = cols.value
        out.append(buf.reshape(-1)[: n * c].reshape(n, c).copy() if n else np.zeros((0, 6), np.float32))
    return out


# --------------------------------------------------------------------------- evaluation accounting
def box_iou(box1, box2, literal=True):
    """box_iou (metrics.py:17-44) in fp32 numpy, the file's operation order.  literal: box1 is [4, N] (what the file
    indexes); otherwise [N, 4].  box2 is [M, 4].  -> [N, M]."""
    b1 = np.asarray(box1, np.float32)
    if not literal:
        b1 = b1.T
    b2 = np.asarray(box2, np.float32).T
    e = np.float32(1e-7)
    inter = np.clip(np.minimum(b1[2][:, None], b2[2]) - np.maximum(b1[0][:, None], b2[0]), 0, None) * \
        np.clip(np.minimum(b1[3][:, None], b2[3]) - np.maximum(b1[1][:, None], b2[1]), 0, None)
    w1, h1 = b1[2] - b1[0], b1[3] - b1[1] + e
    w2, h2 = b2[2] - b2[0], b2[3] - b2[1] + e
    union = (w1[:, None] * h1[:, None]) + (w2 * h2) - inter + e
    return (inter / union).astype(np.float32)


def process_batch(detections, labels, iouv):
    """process_batch (validate.py:71-108) as the YOLOv5 rule it imitates (the file's body cannot run): python loops,
    written independently of the product's vectorised form.  -> bool [n, len(iouv)]."""
    det, lab = np.asarray(detections, np.float32), np.asarray(labels, np.float32)
    n, m = det.shape[0], lab.shape[0]
    correct = np.zeros((n, len(iouv)), dtype=bool)
    if n == 0 or m == 0:
        return correct
    iou = box_iou(lab[:, 1:5], det[:, :4], literal=False)          # [m, n]
    for t, thr in enumerate(iouv):
        pairs = [(float(iou[l, d]), l, d) for l in range(m) for d in range(n) if iou[l, d] >= thr and lab[l, 0] == det[d, 5]]
        # best IoU first; then keep the first pair of every detection (in detection order), then of every label
        pairs.sort(key=lambda x: -x[0])
        by_det = {}
        for p in pairs:
            by_det.setdefault(p[2], p)
        kept = [by_det[d] for d in sorted(by_det)]
        by_lab = {}
        for p in kept:
            by_lab.setdefault(p[1], p)
        for p in by_lab.values():
            correct[p[2], t] = True
    return correct


# --------------------------------------------------------------------------- letterbox
def _lb_taps(n_dst, n_src):
    """OpenCV 8-bit INTER_LINEAR taps: first source index and the two coefficients (x 2048) per destination index."""
    scale = np.float64(n_src) / np.float64(n_dst)
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    s[lo], f[lo] = 0, 0.0
    hi = s >= n_src - 1
    s[hi], f[hi] = n_src - 1, 0.0
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s, a0, a1


def letterbox_pixels(img, new_h, new_w, top, bottom, left, right, pad=114):
    """uint8 [H, W, 3] -> cv2.resize(INTER_LINEAR) restated (fixed point, see csrc/k_misc.hip) + constant border."""
    img = np.asarray(img, np.uint8)
    H0, W0 = img.shape[:2]
    if (new_h, new_w) == (H0, W0):
        res = img
    else:
        xs, ax0, ax1 = _lb_taps(new_w, W0)
        ys, by0, by1 = _lb_taps(new_h, H0)
        x1 = np.minimum(xs + 1, W0 - 1)
        y1 = np.minimum(ys + 1, H0 - 1)
        src = img.astype(np.int64)
        rows0 = src[ys][:, xs] * ax0[None, :, None] + src[ys][:, x1] * ax1[None, :, None]
        rows1 = src[y1][:, xs] * ax0[None, :, None] + src[y1][:, x1] * ax1[None, :, None]
        o = (((by0[:, None, None] * (rows0 >> 4)) >> 16) + ((by1[:, None, None] * (rows1 >> 4)) >> 16) + 2) >> 2
        res = np.clip(o, 0, 255).astype(np.uint8)
    out = np.full((new_h + top + bottom, new_w + left + right, 3), pad, np.uint8)
    out[top:top + new_h, left:left + new_w] = res
    return out


# --------------------------------------------------------------------------- test-time augmentation + tiling
TTA_SCALES = (1.0, 0.83, 0.67)     # the YOLOv5 _forward_augment schedule the reference's `augment=` flag stands for
TTA_FLIPS = (0, 3, 0)              # 3 = left-right (tensor dim 3), 2 = up-down


def scale_img_geometry(h, w, ratio, same_shape=False, gs=32):
    """(resized h, w), (padded h, w) of scale_img, torch_utils.py:275-288 (python float arithmetic, int() truncation)."""
    import math
    s = (int(h * ratio), int(w * ratio))
    if same_shape:
        return s, (h, w)
    return s, tuple(math.ceil(v * ratio / gs) * gs for v in (h, w))


def scale_img(img, ratio=1.0, same_shape=False, gs=32, flip=0):
    """scale_img (torch_utils.py:262-288) of ``img.flip(flip)``: bilinear (align_corners=False) to int(h*ratio) x int(w*ratio),
    zero-origin pad with 0.447 up to the next multiple of ``gs``.  ratio == 1.0 returns the (flipped) image."""
    x = _c(img)
    B, C, H, W = x.shape
    if ratio == 1.0:
        s, p = (H, W), (H, W)
    else:
        s, p = scale_img_geometry(H, W, ratio, same_shape, gs)
    y = np.empty((B, C, p[0], p[1]), np.float32)
    lib = _lib()
    lib.sky_oracle_scale_img.argtypes = [_F, _F] + [ctypes.c_int] * 8 + [ctypes.c_float]
    lib.sky_oracle_scale_img(_p(x), _p(y), B * C, H, W, s[0], s[1], p[0], p[1], int(flip), np.float32(0.447))
    return y


def map_detections(det, scale=1.0, flip=0, img_hw=(0, 0), origins=None):
    """De-scale / un-flip / offset decoded rows [..., (cx, cy, w, h, obj, cls...)]: ``p[..., :4] /= scale``; flip 3:
    ``cx = img_w - cx``; flip 2: ``cy = img_h - cy`` (YOLOv5 _descale_pred); tiles: ``cx += origin_x``, ``cy += origin_y``
    (origins [B, 2] = (y, x) per batch entry).  fp32, one rounding per operation."""
    d = np.array(det, np.float32, copy=True)
    d[..., :4] = d[..., :4] / np.float32(scale)
    if flip == 3:
        d[..., 0] = np.float32(img_hw[1]) - d[..., 0]
    elif flip == 2:
        d[..., 1] = np.float32(img_hw[0]) - d[..., 1]
    if origins is not None:
        o = np.asarray(origins, np.float32)
        d[..., 0] = d[..., 0] + o[:, 1].reshape((-1,) + (1,) * (d.ndim - 2))
        d[..., 1] = d[..., 1] + o[:, 0].reshape((-1,) + (1,) * (d.ndim - 2))
    return d


def tta_rows(n_rows, n_levels=3):
    """Row ranges kept per pass by YOLOv5's _clip_augmented: the full-scale pass drops its last (coarsest) level, the
    smallest pass drops its first (finest) level; rows per level are in the ratio 4^(nl-1) : ... : 1."""
    g = sum(4 ** k for k in range(n_levels))
    return lambda k, n_pass: ((0, n_rows[k] - n_rows[k] // g) if k == 0 else
                              ((n_rows[k] // g) * 4 ** (n_levels - 1), n_rows[k]) if k == n_pass - 1 else (0, n_rows[k]))


def tta_forward(forward, img, gs=32, scales=TTA_SCALES, flips=TTA_FLIPS, clip=False):
    """``forward(x) -> detections [B, N, no]`` applied to every scaled / flipped copy, rows mapped back and concatenated."""
    H, W = img.shape[2:]
    outs = []
    for s, f in zip(scales, flips):
        xi = scale_img(img, s, gs=gs, flip=f)
        outs.append(map_detections(forward(xi), s, f, (H, W)))
    if clip:
        keep = tta_rows([o.shape[1] for o in outs])
        outs = [o[:, slice(*keep(k, len(outs)))] for k, o in enumerate(outs)]
    return np.concatenate(outs, 1)


def tile_origins(h0, w0, tile_h, tile_w, overlap=0.2):
    """Top-left corners (y, x) of the overlapping tiles covering an h0 x w0 frame: step = int(tile * (1 - overlap)),
    the last tile of a row / column is pulled back flush with the border; a frame smaller than a tile gives origin 0."""
    def axis(n, t):
        if n <= t:
            return [0]
        step = max(int(t * (1.0 - overlap)), 1)
        xs = list(range(0, n - t, step)) + [n - t]
        return xs
    return np.array([(y, x) for y in axis(h0, tile_h) for x in axis(w0, tile_w)], np.int32)


def tile_gather(frame_hwc, origins, tile_h, tile_w, pad=114, reverse_channels=False):
    """uint8 [H0, W0, 3] -> uint8 [n, 3, tile_h, tile_w]; outside the frame = ``pad``."""
    f = np.asarray(frame_hwc, np.uint8)
    if reverse_channels:
        f = f[..., ::-1]
    H0, W0 = f.shape[:2]
    out = np.full((len(origins), 3, tile_h, tile_w), pad, np.uint8)
    for i, (y, x) in enumerate(origins):
        h, w = min(tile_h, H0 - y), min(tile_w, W0 - x)
        out[i, :, :h, :w] = f[y:y + h, x:x + w].transpose(2, 0, 1)
    return out
