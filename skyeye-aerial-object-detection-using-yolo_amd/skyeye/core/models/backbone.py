"""Backbones -- drop-in for reference skyeye/core/models/backbone.py."""
import torch.nn as nn

from ._base import NativeModule
from .attention import CombinedAttention
from .blocks import ConvolutionBlock, CSPBlock, FocusBlock, SPPBlock


class Backbone(NativeModule):
    """Four-stage CSP backbone returning [s2, s3, s4]  (reference backbone.py:12-99)."""
    _sky_module = "BACKBONE"

    def __init__(self, base_channels=64, depth_multiple=1.0, width_multiple=1.0):
        super().__init__()

        def scaled_channels(x):
            return max(round(x * width_multiple), 1)

        def scaled_depth(x):
            return max(round(x * depth_multiple), 1)

        c1, c2, c3 = scaled_channels(base_channels), scaled_channels(base_channels * 2), scaled_channels(base_channels * 4)
        c4, c5 = scaled_channels(base_channels * 8), scaled_channels(base_channels * 16)
        self.stage1 = nn.Sequential(FocusBlock(3, c1, kernel_size=3), ConvolutionBlock(c1, c2, 3, stride=2),
                                    CSPBlock(c2, c2, num_blocks=scaled_depth(3)))
        self.stage2 = nn.Sequential(ConvolutionBlock(c2, c3, 3, stride=2), CSPBlock(c3, c3, num_blocks=scaled_depth(9)))
        self.stage3 = nn.Sequential(ConvolutionBlock(c3, c4, 3, stride=2), CSPBlock(c4, c4, num_blocks=scaled_depth(9)),
                                    CombinedAttention(c4))
        self.stage4 = nn.Sequential(ConvolutionBlock(c4, c5, 3, stride=2), CSPBlock(c5, c5, num_blocks=scaled_depth(3)),
                                    SPPBlock(c5, c5))
        self.out_channels = [c3, c4, c5]
        self._cfg = dict(base_channels=base_channels, depth_multiple=float(depth_multiple), width_multiple=float(width_multiple),
                         in_channels=3)

    def _sky_config(self):
        return self._cfg

    def forward(self, x):
        return self._run([x])


class CSPDarknet(Backbone):
    """Alias of Backbone  (reference backbone.py:102-116)."""


class SkyEyeBackbone(nn.Module):
    """Wrapper returning (features, channels)  (reference backbone.py:119-159).

    The reference reports channels [2b, 4b, 8b]*wm although the features have [4b, 8b, 16b]*wm
    (backbone.py:139-143 vs :40-42,99 -- SURVEY App. A D2); this class reports the true ones.
    """

    def __init__(self, base_channels=64, depth_multiple=1.0, width_multiple=1.0):
        super().__init__()
        self.backbone = CSPDarknet(base_channels, depth_multiple, width_multiple)
        self.channels = list(self.backbone.out_channels)

    def forward(self, x):
        return self.backbone(x), self.channels
