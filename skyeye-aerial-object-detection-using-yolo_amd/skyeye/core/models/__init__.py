"""Drop-in for reference skyeye/core/models/__init__.py:5-8 (same re-exports)."""
from .detector import SkyEyeDetector, EnhancedSkyEyeDetector, parse_model, construct_model, load_model
from .blocks import ConvolutionBlock, BottleneckBlock, CSPBlock, SPPBlock, FocusBlock
from .attention import (ChannelAttention, SpatialAttention, CombinedAttention, CrossLayerAttention, TransformerLayer,
                        WindowedSelfAttention)
from .backbone import Backbone, CSPDarknet, SkyEyeBackbone

__all__ = ["SkyEyeDetector", "EnhancedSkyEyeDetector", "parse_model", "construct_model", "load_model", "ConvolutionBlock",
           "BottleneckBlock", "CSPBlock", "SPPBlock", "FocusBlock", "ChannelAttention", "SpatialAttention",
           "CombinedAttention", "CrossLayerAttention", "TransformerLayer", "WindowedSelfAttention", "Backbone",
           "CSPDarknet", "SkyEyeBackbone"]
