"""The detector graph (plain variants: backbone + neck + detection levels + decode) written with torch.nn.functional on the CPU -- the
second CPU comparator of SURVEY 8(d): the build's OWN torch graph, not reference files, timed on the host cores beside the C / OpenMP
port.  Test infrastructure like oracle/: imported by tests/ and by bench.py's cpu_baseline leg only; the product path has no PyTorch
implementation of the math.  Wiring and quirks as in oracle/skyeye_oracle.py (each function cites the reference there); pinned to the
oracle by tests/test_torch_graph.py."""
import torch
import torch.nn.functional as F

ANCHORS = [[[10, 13], [16, 30], [33, 23]], [[30, 61], [62, 45], [59, 119]], [[116, 90], [156, 198], [373, 326]]]      # detector.py:39-43


def conv_block(P, pre, x, k, stride=1):
    """ConvolutionBlock (blocks.py:10-37): SiLU(BN_eval(conv))."""
    y = F.conv2d(x, P[pre + "conv.weight"], None, stride, k // 2)
    y = F.batch_norm(y, P[pre + "bn.running_mean"], P[pre + "bn.running_var"], P[pre + "bn.weight"], P[pre + "bn.bias"], False, 0.0, 1e-5)
    return F.silu(y)


def csp(P, pre, x):
    """CSPBlock (blocks.py:93-123) with BottleneckBlock (blocks.py:69-90)."""
    y1 = conv_block(P, pre + "cv1.", x, 1)
    j = 0
    while f"{pre}bottlenecks.{j}.cv1.conv.weight" in P:
        b = f"{pre}bottlenecks.{j}."
        y1 = y1 + conv_block(P, b + "cv2.", conv_block(P, b + "cv1.", y1, 1), 3)
        j += 1
    return conv_block(P, pre + "cv3.", torch.cat((y1, conv_block(P, pre + "cv2.", x, 1)), 1), 1)


def backbone(P, pre, x):
    """Backbone.forward (backbone.py:82-99)."""
    s = torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)       # FocusBlock, blocks.py:176-181
    s = conv_block(P, pre + "stage1.0.conv.", s, 3)
    s1 = csp(P, pre + "stage1.2.", conv_block(P, pre + "stage1.1.", s, 3, 2))
    s2 = csp(P, pre + "stage2.1.", conv_block(P, pre + "stage2.0.", s1, 3, 2))
    s3 = csp(P, pre + "stage3.1.", conv_block(P, pre + "stage3.0.", s2, 3, 2))
    ca = pre + "stage3.2.channel_attention."                                                               # attention.py:11-60
    mlp = lambda v: F.linear(F.relu(F.linear(v, P[ca + "shared_mlp.0.weight"])), P[ca + "shared_mlp.2.weight"])
    s3 = s3 * torch.sigmoid(mlp(s3.mean((2, 3))) + mlp(s3.amax((2, 3))))[:, :, None, None]
    sa = torch.cat([s3.mean(1, keepdim=True), s3.amax(1, keepdim=True)], 1)                               # attention.py:63-98
    s3 = s3 * torch.sigmoid(F.conv2d(sa, P[pre + "stage3.2.spatial_attention.conv.weight"], None, 1, 3))
    s4 = csp(P, pre + "stage4.1.", conv_block(P, pre + "stage4.0.", s3, 3, 2))
    y = conv_block(P, pre + "stage4.2.cv1.", s4, 1)                                                        # SPPBlock, blocks.py:126-149
    s4 = conv_block(P, pre + "stage4.2.cv2.", torch.cat([y] + [F.max_pool2d(y, k, 1, k // 2) for k in (5, 9, 13)], 1), 1)
    return s2, s3, s4


def neck(P, pre, p3, p4, p5):
    """FeatureNeck.forward (detector.py:197-231), quirks kept (raw p4 into lateral_conv4, raw p5 into the last concat)."""
    p5_td = conv_block(P, pre + "lateral_conv5.", p5, 1)
    p4_td = conv_block(P, pre + "lateral_conv4.", p4, 1)
    p4p = csp(P, pre + "fpn_conv4.", torch.cat([F.interpolate(p5_td, size=p4.shape[2:], mode="nearest"), p4], 1))
    p3p = csp(P, pre + "fpn_conv3.", torch.cat([F.interpolate(p4_td, size=p3.shape[2:], mode="nearest"), p3], 1))
    p4o = csp(P, pre + "pan_conv4.", torch.cat([conv_block(P, pre + "downsample3.", p3p, 3, 2), p4p], 1))
    p5o = csp(P, pre + "pan_conv5.", torch.cat([conv_block(P, pre + "downsample4.", p4o, 3, 2), p5], 1))
    return p3p, p4o, p5o


@torch.no_grad()
def detector_forward(P, x, nc):
    """SkyEyeDetector.forward in eval mode (detector.py:300-324) + process_detections (detector.py:88-145, anchor x stride quirk kept)."""
    feats = neck(P, "neck.", *backbone(P, "backbone.backbone.", x))
    no, dets = nc + 5, []
    for i, f in enumerate(feats):
        y = F.conv2d(f, P[f"detection_head.detection_layers.{i}.weight"], P[f"detection_head.detection_layers.{i}.bias"])
        B, _, gh, gw = y.shape
        out = y.view(B, 3, no, gh, gw).permute(0, 1, 3, 4, 2).contiguous()
        stride = float(max(x.shape[2] / gh, x.shape[3] / gw))
        yv, xv = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, 1, gh, gw, 2).float()
        ag = torch.tensor(ANCHORS[i], dtype=torch.float32).view(1, 3, 1, 1, 2) * stride
        s = torch.sigmoid(out)
        s[..., 0:2] = (s[..., 0:2] * 2.0 - 0.5 + grid) * stride
        s[..., 2:4] = (s[..., 2:4] * 2.0) ** 2 * ag
        dets.append(s.view(B, -1, no))
    return torch.cat(dets, 1)
