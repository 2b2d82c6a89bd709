"""Case tables shared by the golden generator and the parity tests.

Each case names a reference class (file:line in the docstrings of
``make_golden.py``), its constructor arguments, the seeded inputs and the seed.
Only *outputs* are stored in the fixtures; weights and inputs are regenerated
from ``seeded.py`` on both sides.
"""

# model definitions are the build's (the reference YAMLs are empty)
MODELS = {
    "skyeye_s": dict(base_channels=64, depth_multiple=0.33, width_multiple=0.50, nc=10),
    "skyeye_m": dict(base_channels=64, depth_multiple=0.67, width_multiple=0.75, nc=10),
    "skyeye_l": dict(base_channels=64, depth_multiple=1.00, width_multiple=1.00, nc=10),
}

# Weight seed per detector variant.  BatchNorm running statistics of the full detectors are NOT taken from
# seeded.py: make_golden.py calibrates them once per variant on a 128x128 batch (like training would) and stores
# them in bn_calib.npz, otherwise a 100-layer random network is chaotic and every comparison degenerates into a
# test of overflow.  Key in bn_calib.npz: "<variant>:<state-dict name>".
WSEED = {"skyeye_s": 101, "skyeye_m": 103, "skyeye_l": 104, "skyeye_s_enh": 109, "skyeye_s_ha": 110}


def variant_of(case):
    return case["model"] + ("_enh" if case.get("enhanced") else "") + ("_ha" if case.get("head_attention") else "")


# ---- per-block cases (SURVEY 8c fixture list item 1) -----------------------
# kind -> reference class; args -> constructor kwargs; inputs -> {name: (shape, lo, hi)}
BLOCK_CASES = [
    dict(name="conv_k1", kind="ConvolutionBlock", args=dict(in_channels=32, out_channels=64, kernel_size=1, stride=1),
         inputs=dict(x=((2, 32, 12, 10), -1.0, 1.0)), seed=11),
    dict(name="conv_k3s1", kind="ConvolutionBlock", args=dict(in_channels=32, out_channels=32, kernel_size=3, stride=1),
         inputs=dict(x=((2, 32, 12, 10), -1.0, 1.0)), seed=12),
    dict(name="conv_k3s2", kind="ConvolutionBlock", args=dict(in_channels=32, out_channels=64, kernel_size=3, stride=2),
         inputs=dict(x=((2, 32, 12, 10), -1.0, 1.0)), seed=13),
    dict(name="conv_k3s2_odd", kind="ConvolutionBlock", args=dict(in_channels=48, out_channels=40, kernel_size=3, stride=2),
         inputs=dict(x=((1, 48, 11, 9), -1.0, 1.0)), seed=14),
    dict(name="conv_k1_noact", kind="ConvolutionBlock",
         args=dict(in_channels=64, out_channels=32, kernel_size=1, stride=1, activation=False),
         inputs=dict(x=((1, 64, 5, 7), -1.0, 1.0)), seed=15),
    dict(name="focus", kind="FocusBlock", args=dict(in_channels=3, out_channels=32, kernel_size=3),
         inputs=dict(x=((2, 3, 16, 24), 0.0, 1.0)), seed=16),
    dict(name="bottleneck", kind="BottleneckBlock", args=dict(in_channels=64, out_channels=64, shortcut=True, expansion=1.0),
         inputs=dict(x=((2, 64, 8, 8), -1.0, 1.0)), seed=17),
    dict(name="bottleneck_noshortcut", kind="BottleneckBlock",
         args=dict(in_channels=64, out_channels=32, shortcut=True, expansion=0.5),
         inputs=dict(x=((1, 64, 6, 8), -1.0, 1.0)), seed=18),
    dict(name="csp_n2", kind="CSPBlock", args=dict(in_channels=64, out_channels=64, num_blocks=2),
         inputs=dict(x=((2, 64, 8, 8), -1.0, 1.0)), seed=19),
    dict(name="csp_neck", kind="CSPBlock", args=dict(in_channels=192, out_channels=128, num_blocks=3),
         inputs=dict(x=((1, 192, 6, 10), -1.0, 1.0)), seed=20),
    dict(name="spp", kind="SPPBlock", args=dict(in_channels=64, out_channels=64),
         inputs=dict(x=((2, 64, 10, 7), -1.0, 1.0)), seed=21),
    dict(name="spp_big", kind="SPPBlock", args=dict(in_channels=128, out_channels=128),
         inputs=dict(x=((1, 128, 20, 20), -1.0, 1.0)), seed=22),
    dict(name="channel_attention", kind="ChannelAttention", args=dict(channels=64),
         inputs=dict(x=((2, 64, 9, 11), -1.0, 2.0)), seed=23),
    dict(name="spatial_attention", kind="SpatialAttention", args=dict(),
         inputs=dict(x=((2, 64, 9, 11), -1.0, 2.0)), seed=24),
    dict(name="combined_attention", kind="CombinedAttention", args=dict(channels=128),
         inputs=dict(x=((2, 128, 10, 6), -1.0, 2.0)), seed=25),
    dict(name="backbone_s", kind="Backbone", args=dict(base_channels=64, depth_multiple=0.33, width_multiple=0.5),
         inputs=dict(x=((2, 3, 64, 96), 0.0, 1.0)), seed=26),
    dict(name="neck_s", kind="FeatureNeck", args=dict(in_channels=[128, 256, 512], width_multiple=1.0),
         inputs=dict(p3=((2, 128, 8, 12), -1.0, 1.0), p4=((2, 256, 4, 6), -1.0, 1.0), p5=((2, 512, 2, 3), -1.0, 1.0)),
         seed=27),
    dict(name="head_decode", kind="DetectionHead", args=dict(num_classes=10, anchors=None, channels=[128, 256, 512]),
         inputs=dict(p3=((2, 128, 8, 12), -1.0, 1.0), p4=((2, 256, 4, 6), -1.0, 1.0), p5=((2, 512, 2, 3), -1.0, 1.0)),
         input_shape=(64, 96), seed=28),
    dict(name="head_decode_nc1", kind="DetectionHead",
         args=dict(num_classes=1, anchors=[[[8, 9], [20, 17]], [[40, 33], [70, 90]]], channels=[64, 96]),
         inputs=dict(p3=((1, 64, 6, 5), -1.0, 1.0), p4=((1, 96, 3, 3), -1.0, 1.0)),
         input_shape=(48, 40), seed=29),
    # attention modules of config 3 (pinned individually)
    dict(name="cla_equal", kind="CrossLayerAttention",
         args=dict(query_channels=64, key_channels=64, region_size=2, heads=4),
         inputs=dict(q=((2, 64, 8, 10), -1.0, 1.0), k=((2, 64, 4, 5), -1.0, 1.0)), seed=30),
    dict(name="cla_d4", kind="CrossLayerAttentionD4",
         args=dict(query_channels=64, key_channels=128, region_size=2, heads=4),
         inputs=dict(q=((2, 64, 8, 10), -1.0, 1.0), k=((2, 128, 4, 5), -1.0, 1.0)), seed=31),
    dict(name="transformer", kind="TransformerLayer", args=dict(dim=64, num_heads=4),
         inputs=dict(x=((2, 64, 5, 6), -1.0, 1.0)), seed=32),
    dict(name="windowed_attention", kind="WindowedSelfAttention", args=dict(dim=64, window_size=4, num_heads=4),
         inputs=dict(x=((6, 16, 64), -1.0, 1.0)), seed=33),
    dict(name="windowed_attention_mask", kind="WindowedSelfAttention", args=dict(dim=32, window_size=2, num_heads=2),
         inputs=dict(x=((6, 4, 32), -1.0, 1.0), mask=((3, 4, 4), -2.0, 0.0)), seed=34),
]

# ---- full-graph cases ----------------------------
# small: full tensors stored.  big: sha256 + sampled rows stored.
DETECTOR_CASES = [
    dict(name="s_64x64", model="skyeye_s", batch=2, hw=(64, 64), seed=101, store="full"),
    dict(name="s_160x192", model="skyeye_s", batch=2, hw=(160, 192), seed=102, store="full"),
    dict(name="m_128x96", model="skyeye_m", batch=1, hw=(128, 96), seed=103, store="full"),
    dict(name="l_128x128", model="skyeye_l", batch=2, hw=(128, 128), seed=104, store="full"),
    dict(name="l_96x160", model="skyeye_l", batch=1, hw=(96, 160), seed=105, store="full"),
    dict(name="s_640", model="skyeye_s", batch=1, hw=(640, 640), seed=106, store="sampled"),
    dict(name="s_1280", model="skyeye_s", batch=1, hw=(1280, 1280), seed=107, store="sampled"),
    dict(name="l_640", model="skyeye_l", batch=1, hw=(640, 640), seed=108, store="sampled"),
    dict(name="enh_s_128x96", model="skyeye_s", batch=2, hw=(128, 96), seed=109, store="full", enhanced=True),
    dict(name="l_1280", model="skyeye_l", batch=1, hw=(1280, 1280), seed=113, store="sampled"),     # config 4's per-GPU graph at its own size
]
N_SAMPLED_ROWS = 4096

# ---- "transformer prediction heads" (SURVEY App. A, D5): build-defined call site of the reference's TransformerLayer /
# WindowedSelfAttention modules; fixtures in detectors_ha.npz, BatchNorm calibration in bn_calib_ha.npz
HA_CASES = [
    dict(name="ha_s_128x128", model="skyeye_s", batch=2, hw=(128, 128), seed=111, store="full", head_attention=True),
    dict(name="ha_s_128x256", model="skyeye_s", batch=1, hw=(128, 256), seed=112, store="full", head_attention=True),
    dict(name="ha_s_1280", model="skyeye_s", batch=1, hw=(1280, 1280), seed=114, store="sampled", head_attention=True),   # config 3 at its own size
]

# ---- NMS wrapper cases -----------------------------------
# predictions are synthetic [B, N, nc+5] tensors built by tests/golden/nms_inputs.py
NMS_CASES = [
    dict(name="nc1", nc=1, batch=2, n=600, seed=201, kwargs=dict(conf_threshold=0.25, iou_threshold=0.45)),
    dict(name="single_label", nc=10, batch=3, n=1500, seed=202, kwargs=dict(conf_threshold=0.25, iou_threshold=0.45)),
    dict(name="multi_label", nc=6, batch=2, n=800, seed=203,
         kwargs=dict(conf_threshold=0.3, iou_threshold=0.5, multi_label=True)),
    # literal quirk (D9): for nc>1 the filter compares column 5 = cls_conf with the class ids,
    # so only rows whose confidence is exactly 1.0 survive classes=[1]
    dict(name="classes_filter", nc=10, batch=2, n=1200, seed=204, ties=True,
         kwargs=dict(conf_threshold=0.25, iou_threshold=0.45, classes=[1, 3, 7])),
    dict(name="nc1_classes", nc=1, batch=2, n=400, seed=211,
         kwargs=dict(conf_threshold=0.25, iou_threshold=0.45, classes=[0])),
    dict(name="agnostic", nc=10, batch=2, n=1200, seed=205,
         kwargs=dict(conf_threshold=0.25, iou_threshold=0.45, agnostic=True)),
    dict(name="max_det_small", nc=10, batch=2, n=2000, seed=206,
         kwargs=dict(conf_threshold=0.05, iou_threshold=0.6, max_detections=50)),
    dict(name="over_cap", nc=3, batch=1, n=33000, seed=207, kwargs=dict(conf_threshold=0.001, iou_threshold=0.6)),
    dict(name="ties", nc=4, batch=2, n=500, seed=208, kwargs=dict(conf_threshold=0.25, iou_threshold=0.45), ties=True),
    dict(name="empty", nc=10, batch=2, n=300, seed=209, kwargs=dict(conf_threshold=0.999, iou_threshold=0.45)),
    dict(name="corrected_mode_like", nc=10, batch=2, n=1500, seed=210,
         kwargs=dict(conf_threshold=0.1, iou_threshold=0.3)),
]


# scale_img (torch_utils.py:262-288) cases of the test-time-augmentation front end:
# name -> (B, H, W, ratio, flip dim of the caller (0 none / 2 / 3), same_shape, gs).  Inputs: seeded_input("tta." + name).
TTA_CASES = {
    "r083": (2, 64, 96, 0.83, 0, False, 32),
    "r067_lr": (1, 64, 96, 0.67, 3, False, 32),
    "r083_ud": (1, 96, 160, 0.83, 2, False, 32),
    "r050_same": (1, 128, 128, 0.5, 0, True, 32),
    "r067_gs64": (1, 160, 224, 0.67, 0, False, 64),
    "r100": (1, 32, 64, 1.0, 0, False, 32),
    "r083_big": (1, 640, 640, 0.83, 3, False, 32),
}
