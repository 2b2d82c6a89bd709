"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

The detection forward pass of ``oracle/skyeye_oracle.py`` evaluated in FLOAT64 (numpy only: convolutions as im2col + BLAS dgemm):
the same restatement of the reference (file:line citations there, function by function in the same order), every operand and every
intermediate in double.  It is not a parity oracle -- the reference computes in fp32 -- but the yardstick that APPORTIONS an fp32
difference: with e = |engine - fixture| above 1e-4 on a case, |f64 - fixture| and |f64 - engine| say whose rounding it is
(tools/f64_error_budget.py, tests/test_f64_error_budget.py; round-3 review item 2).  Covers what the three exceptions of
tests/test_gpu_detector.py need: the plain detector (skyeye_s / skyeye_l) and the Enhanced detector's cross-layer attention.

Parity status: the f64 graph is pinned INDIRECTLY -- rounded to fp32 it must sit within the fixtures' own fp32 noise of every reference
fixture it covers (tests/test_f64_error_budget.py checks that against detectors_full / detectors_sampled)."""
import numpy as np

F = np.float64


def conv2d(x, w, bias=None, stride=1, pad=None):
    """nn.Conv2d forward (blocks.py:31, detector.py:56-59, attention.py:79,167-170) as w[Co, Ci*K*K] @ im2col(x), row bands of the output
    so that the column matrix stays below ~0.5 GB."""
    x = np.asarray(x, F)
    w = np.asarray(w, F)
    B, Ci, H, W = x.shape
    Co, Ci2, K, K2 = w.shape
    assert Ci == Ci2 and K == K2
    pad = K // 2 if pad is None else pad
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    y = np.empty((B, Co, Ho, Wo), F)
    wm = w.reshape(Co, Ci * K * K)
    if K == 1 and stride == 1 and pad == 0:
        for b in range(B):
            y[b] = (wm @ x[b].reshape(Ci, H * W)).reshape(Co, Ho, Wo)
    else:
        xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad))) if pad else x
        band = max(1, int(6e7 // max(1, Ci * K * K * Wo)))
        for b in range(B):
            win = np.lib.stride_tricks.sliding_window_view(xp[b], (K, K), axis=(1, 2))[:, ::stride, ::stride]       # [Ci, Ho, Wo, K, K]
            for r0 in range(0, Ho, band):
                r1 = min(Ho, r0 + band)
                cols = np.ascontiguousarray(win[:, r0:r1].transpose(0, 3, 4, 1, 2)).reshape(Ci * K * K, (r1 - r0) * Wo)
                y[b, :, r0:r1] = (wm @ cols).reshape(Co, r1 - r0, Wo)
    if bias is not None:
        y += np.asarray(bias, F)[None, :, None, None]
    return y


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def maxpool2d(x, k):
    """nn.MaxPool2d(k, stride=1, padding=k//2) with -inf padding (blocks.py:142-144)."""
    p = k // 2
    xp = np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)), constant_values=-np.inf)
    return np.lib.stride_tricks.sliding_window_view(xp, (k, k), axis=(2, 3)).max(axis=(-1, -2))


def upsample_nearest(x, size):
    """F.interpolate(size=..., mode='nearest') (detector.py:214,218): src = floor(dst * in / out)."""
    H, W = x.shape[2:]
    iy = (np.arange(size[0]) * H) // size[0]
    ix = (np.arange(size[1]) * W) // size[1]
    return x[:, :, iy][:, :, :, ix]


def bilinear(x, size):
    """F.interpolate(size=..., mode='bilinear', align_corners=False) (attention.py:211-212)."""
    H, W = x.shape[2:]

    def taps(n_dst, n_src):
        s = np.maximum((np.arange(n_dst, dtype=F) + 0.5) * (n_src / n_dst) - 0.5, 0.0)
        i0 = np.minimum(np.floor(s).astype(np.int64), n_src - 1)
        i1 = np.minimum(i0 + 1, n_src - 1)
        return i0, i1, s - i0

    y0, y1, fy = taps(size[0], H)
    x0, x1, fx = taps(size[1], W)
    top = x[:, :, y0][:, :, :, x0] * (1 - fx) + x[:, :, y0][:, :, :, x1] * fx
    bot = x[:, :, y1][:, :, :, x0] * (1 - fx) + x[:, :, y1][:, :, :, x1] * fx
    return top * (1 - fy)[None, None, :, None] + bot * fy[None, None, :, None]


def linear(x, w, bias=None):
    y = x @ np.asarray(w, F).T
    return y if bias is None else y + np.asarray(bias, F)


# --------------------------------------------------------------------------- blocks.py
def conv_block(P, pre, x, k, stride=1, act=True):
    """ConvolutionBlock.forward: act(bn(conv(x)))  (blocks.py:10-37); BN eps = 1e-5 (eval), SiLU."""
    y = conv2d(x, P[pre + "conv.weight"], None, stride, k // 2)
    g, b = np.asarray(P[pre + "bn.weight"], F), np.asarray(P[pre + "bn.bias"], F)
    mu, var = np.asarray(P[pre + "bn.running_mean"], F), np.asarray(P[pre + "bn.running_var"], F)
    y = (y - mu[None, :, None, None]) / np.sqrt(var + 1e-5)[None, :, None, None] * g[None, :, None, None] + b[None, :, None, None]
    return y * sigmoid(y) if act else y


def bottleneck(P, pre, x, shortcut=True):
    y = conv_block(P, pre + "cv2.", conv_block(P, pre + "cv1.", x, 1), 3)
    use = shortcut and P[pre + "cv1.conv.weight"].shape[1] == P[pre + "cv2.conv.weight"].shape[0]
    return x + y if use else y


def csp(P, pre, x):
    y1 = conv_block(P, pre + "cv1.", x, 1)
    j = 0
    while f"{pre}bottlenecks.{j}.cv1.conv.weight" in P:
        y1 = bottleneck(P, f"{pre}bottlenecks.{j}.", y1)
        j += 1
    y2 = conv_block(P, pre + "cv2.", x, 1)
    return conv_block(P, pre + "cv3.", np.concatenate((y1, y2), 1), 1)


def spp(P, pre, x, kernel_sizes=(5, 9, 13)):
    y = conv_block(P, pre + "cv1.", x, 1)
    return conv_block(P, pre + "cv2.", np.concatenate([y] + [maxpool2d(y, k) for k in kernel_sizes], 1), 1)


def focus(P, pre, x):
    patches = [x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]]
    return conv_block(P, pre + "conv.", np.concatenate(patches, 1), int(P[pre + "conv.conv.weight"].shape[-1]))


# --------------------------------------------------------------------------- attention.py
def channel_attention(P, pre, x):
    avg, mx = x.mean(axis=(2, 3)), x.max(axis=(2, 3))
    w0, w2 = P[pre + "shared_mlp.0.weight"], P[pre + "shared_mlp.2.weight"]

    def mlp(v):
        return linear(np.maximum(linear(v, w0), 0.0), w2)

    return x * sigmoid(mlp(avg) + mlp(mx))[:, :, None, None]


def spatial_attention(P, pre, x):
    a = sigmoid(conv2d(np.concatenate([x.mean(axis=1, keepdims=True), x.max(axis=1, keepdims=True)], 1), P[pre + "conv.weight"], None, 1, 3))
    return x * a


def combined_attention(P, pre, x):
    return spatial_attention(P, pre + "spatial_attention.", channel_attention(P, pre + "channel_attention.", x))


def cross_layer_attention(P, pre, query, key, heads=4, region_size=2):
    """CrossLayerAttention.forward (attention.py:174-241) in the closed form of oracle/skyeye_oracle.py."""
    q = conv2d(query, P[pre + "query_projection.weight"], P[pre + "query_projection.bias"], 1, 0)
    k = conv2d(key, P[pre + "key_projection.weight"], P[pre + "key_projection.bias"], 1, 0)
    v = conv2d(key, P[pre + "value_projection.weight"], P[pre + "value_projection.bias"], 1, 0)
    B, Cq, H, W = q.shape
    k, v = bilinear(k, (H, W)), bilinear(v, (H, W))
    d = Cq // heads
    s = (q.reshape(B, heads, d, H, W) * k.reshape(B, heads, d, H, W)).sum(2) / np.sqrt(F(query.shape[1]))
    e = np.exp(s - s.max(axis=2, keepdims=True))
    a = e / e.sum(axis=2, keepdims=True)                                                               # softmax over image rows
    out = (a[:, :, None] * v.reshape(B, heads, d, H, W)) * F(region_size * region_size)
    return conv2d(out.reshape(B, Cq, H, W), P[pre + "output_projection.weight"], P[pre + "output_projection.bias"], 1, 0)


# --------------------------------------------------------------------------- backbone.py / detector.py
def backbone(P, pre, x):
    s = focus(P, pre + "stage1.0.", x)
    s = conv_block(P, pre + "stage1.1.", s, 3, 2)
    s1 = csp(P, pre + "stage1.2.", s)
    s2 = csp(P, pre + "stage2.1.", conv_block(P, pre + "stage2.0.", s1, 3, 2))
    s3 = csp(P, pre + "stage3.1.", conv_block(P, pre + "stage3.0.", s2, 3, 2))
    s3 = combined_attention(P, pre + "stage3.2.", s3)
    s4 = csp(P, pre + "stage4.1.", conv_block(P, pre + "stage4.0.", s3, 3, 2))
    return [s2, s3, spp(P, pre + "stage4.2.", s4)]


def feature_neck(P, pre, feats):
    p3, p4, p5 = feats
    p5_td = conv_block(P, pre + "lateral_conv5.", p5, 1)
    p4_td = conv_block(P, pre + "lateral_conv4.", p4, 1)
    p4_processed = csp(P, pre + "fpn_conv4.", np.concatenate([upsample_nearest(p5_td, p4.shape[2:]), p4], 1))
    p3_processed = csp(P, pre + "fpn_conv3.", np.concatenate([upsample_nearest(p4_td, p3.shape[2:]), p3], 1))
    p4_out = csp(P, pre + "pan_conv4.", np.concatenate([conv_block(P, pre + "downsample3.", p3_processed, 3, 2), p4_processed], 1))
    p5_out = csp(P, pre + "pan_conv5.", np.concatenate([conv_block(P, pre + "downsample4.", p4_out, 3, 2), p5], 1))
    return [p3_processed, p4_out, p5_out]


DEFAULT_ANCHORS = [[[10, 13], [16, 30], [33, 23]], [[30, 61], [62, 45], [59, 119]], [[116, 90], [156, 198], [373, 326]]]


def detector_forward(P, x, nc, anchors=None, enhanced=False):
    """-> (det [B, N, nc + 5], [raw levels [B, na, gh, gw, nc + 5]]) in float64; ``x``: float64 frames in [0, 1] (the caller divides
    the uint8 frames by 255 in double)."""
    anchors = DEFAULT_ANCHORS if anchors is None else anchors
    x = np.asarray(x, F)
    neck = feature_neck(P, "neck.", backbone(P, "backbone.backbone.", x))
    if enhanced:
        p3, p4, p5 = neck
        p4e = cross_layer_attention(P, "cross_attention_p5_p4.", p4, p5) + p4
        p3e = cross_layer_attention(P, "cross_attention_p4_p3.", p3, p4e) + p3
        neck = [p3e, p4e, p5]
    na, no = len(anchors[0]), nc + 5
    raw, dets = [], []
    for i, f in enumerate(neck):
        y = conv2d(f, P[f"detection_head.detection_layers.{i}.weight"], P[f"detection_head.detection_layers.{i}.bias"], 1, 0)
        B, _, gh, gw = y.shape
        out = np.ascontiguousarray(y.reshape(B, na, no, gh, gw).transpose(0, 1, 3, 4, 2))
        raw.append(out)
        stride = F(max(x.shape[2] / gh, x.shape[3] / gw))
        yv, xv = np.meshgrid(np.arange(gh), np.arange(gw), indexing="ij")
        grid = np.stack((xv, yv), 2).reshape(1, 1, gh, gw, 2).astype(F)
        anchor_grid = np.asarray(anchors[i], F).reshape(1, na, 1, 1, 2) * stride
        s = sigmoid(out)
        s[..., 0:2] = (s[..., 0:2] * 2 - 0.5 + grid) * stride
        s[..., 2:4] = (s[..., 2:4] * 2) ** 2 * anchor_grid
        dets.append(s.reshape(B, -1, no))
    return np.concatenate(dets, 1), raw
