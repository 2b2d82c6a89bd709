"""Attention modules -- drop-in for reference skyeye/core/models/attention.py."""
import torch
import torch.nn as nn

from ._base import Conv2dParams, LayerNormParams, LinearParams, Marker, NativeModule, _Holder


class ChannelAttention(NativeModule):
    """x * sigmoid(mlp(avgpool(x)) + mlp(maxpool(x)))  (reference attention.py:11-60)."""
    _sky_module = "CHANNEL_ATTENTION"

    def __init__(self, channels, reduction_ratio=16):
        super().__init__()
        reduced_channels = max(channels // reduction_ratio, 1)
        self.avg_pool = Marker("AdaptiveAvgPool2d(1)")
        self.max_pool = Marker("AdaptiveMaxPool2d(1)")
        self.shared_mlp = nn.Sequential(LinearParams(channels, reduced_channels, bias=False), Marker("ReLU"),
                                        LinearParams(reduced_channels, channels, bias=False))
        self.sigmoid = Marker("Sigmoid")
        self._cfg = dict(c_in=channels, reduction_ratio=reduction_ratio)

    def _sky_config(self):
        return self._cfg


class SpatialAttention(NativeModule):
    """x * sigmoid(conv7x7(cat(mean_c(x), max_c(x))))  (reference attention.py:63-98)."""
    _sky_module = "SPATIAL_ATTENTION"

    def __init__(self, kernel_size=7):
        super().__init__()
        if kernel_size != 7:
            raise NotImplementedError("the reference only uses kernel_size=7 (attention.py:68)")
        self.conv = Conv2dParams(2, 1, kernel_size, padding=kernel_size // 2, bias=False)
        self.sigmoid = Marker("Sigmoid")
        self._cfg = dict(c_in=0)

    def _sky_config(self):
        return self._cfg

    def _run(self, inputs, extra_cfg=None):
        return super()._run(inputs, dict(c_in=int(inputs[0].shape[1])))


class CombinedAttention(NativeModule):
    """ChannelAttention then SpatialAttention (CBAM)  (reference attention.py:101-130)."""
    _sky_module = "COMBINED_ATTENTION"

    def __init__(self, channels, reduction_ratio=16):
        super().__init__()
        self.channel_attention = ChannelAttention(channels, reduction_ratio)
        self.spatial_attention = SpatialAttention()
        self._cfg = dict(c_in=channels, reduction_ratio=reduction_ratio)

    def _sky_config(self):
        return self._cfg


class CrossLayerAttention(NativeModule):
    """reference attention.py:133-241 (column-softmax closed form)."""
    _sky_module = "CROSS_LAYER_ATTENTION"

    def __init__(self, query_channels, key_channels, value_channels=None, region_size=2, output_channels=None, heads=4,
                 project_key_to_query=False):
        super().__init__()
        if value_channels is None:
            value_channels = key_channels
        if output_channels is None:
            output_channels = query_channels
        if value_channels != key_channels:
            raise NotImplementedError("value is always the key tensor on the detector's path (detector.py:488-489)")
        self.heads, self.region_size = heads, region_size
        self.query_channels, self.key_channels, self.value_channels = query_channels, key_channels, value_channels
        # D4: as written the module needs equal per-head widths; the Enhanced detector wires
        # key_channels != query_channels, for which key/value projections map key_channels -> query_channels.
        kv_out = query_channels if (project_key_to_query or key_channels != query_channels) else key_channels
        self.query_projection = Conv2dParams(query_channels, query_channels, 1, bias=True)
        self.key_projection = Conv2dParams(key_channels, kv_out, 1, bias=True)
        self.value_projection = Conv2dParams(value_channels, kv_out, 1, bias=True)
        self.output_projection = Conv2dParams(kv_out, output_channels, 1, bias=True)
        self.attention_softmax = Marker("Softmax(dim=3)")
        self._cfg = dict(c_in=query_channels, c_out=output_channels, key_channels=key_channels, heads=heads,
                         region_size=region_size)

    def _sky_config(self):
        return self._cfg

    def forward(self, query, key, value=None):
        if value is not None and value is not key:
            raise NotImplementedError("value defaults to key on the detector's path")
        return self._run([query, key])[0]


class _MHAParams(_Holder):
    """Stands for nn.MultiheadAttention(dim, heads) (packed in_proj, out_proj), reference attention.py:264."""

    def __init__(self, dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads = dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = LinearParams(dim, dim, bias=True)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class TransformerLayer(NativeModule):
    """pre-LN encoder layer over the H*W tokens of a feature map  (reference attention.py:244-309), eval mode."""
    _sky_module = "TRANSFORMER_LAYER"

    def __init__(self, dim, num_heads, feedforward_dim=None, dropout=0.1):
        super().__init__()
        if feedforward_dim is None:
            feedforward_dim = dim * 4
        self.self_attn = _MHAParams(dim, num_heads)
        self.norm1 = LayerNormParams(dim)
        self.norm2 = LayerNormParams(dim)
        self.feedforward = nn.Sequential(LinearParams(dim, feedforward_dim), Marker("ReLU"), Marker("Dropout"),
                                         LinearParams(feedforward_dim, dim), Marker("Dropout"))
        self.dropout = Marker("Dropout")
        self._cfg = dict(c_in=dim, c_out=feedforward_dim, heads=num_heads)

    def _sky_config(self):
        return self._cfg


class WindowedSelfAttention(NativeModule):
    """Swin-style window attention with relative position bias  (reference attention.py:312-399)."""
    _sky_module = "WINDOWED_ATTENTION"

    def __init__(self, dim, window_size, num_heads):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = LinearParams(dim, dim * 3)
        self.proj = LinearParams(dim, dim)
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) * (2 * window_size - 1), num_heads))
        coords = torch.stack(torch.meshgrid(torch.arange(window_size), torch.arange(window_size), indexing="ij"))
        cf = torch.flatten(coords, 1)
        rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += window_size - 1
        rel[:, :, 1] += window_size - 1
        rel[:, :, 0] *= 2 * window_size - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self._cfg = dict(c_in=dim, window_size=window_size, heads=num_heads)

    def _sky_config(self):
        return self._cfg

    def forward(self, x, mask=None):
        return self._run([x] if mask is None else [x, mask])[0]
