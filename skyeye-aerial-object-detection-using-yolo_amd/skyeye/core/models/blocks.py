"""Building blocks -- drop-in for reference skyeye/core/models/blocks.py (same classes, arguments, state-dict keys).

Each ``forward`` is a single call into the HIP engine; see csrc/engine.cpp for how a block becomes launches.
"""
import torch.nn as nn

from ._base import BatchNormParams, Conv2dParams, Marker, NativeModule


class ConvolutionBlock(NativeModule):
    """conv(bias=False) + BatchNorm(eval) + SiLU  (reference blocks.py:10-41).  BN is folded into the packed
    weights and SiLU runs in the GEMM epilogue, so ``fused_forward`` (blocks.py:39-41) is the same call."""
    _sky_module = "CONV_BLOCK"

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, groups=1, activation=True):
        super().__init__()
        if groups != 1:
            raise NotImplementedError("groups != 1 is not on the detector's path (DepthwiseSeparableConv is unused, blocks.py:44-66)")
        if padding is None:
            padding = kernel_size // 2
        if padding != kernel_size // 2:
            raise NotImplementedError("only auto-padding k//2 (blocks.py:28-29) is supported")
        self.conv = Conv2dParams(in_channels, out_channels, kernel_size, stride, padding, bias=False)
        self.bn = BatchNormParams(out_channels)
        self.act = Marker("SiLU" if activation else "Identity")
        self._cfg = dict(c_in=in_channels, c_out=out_channels, kernel_size=kernel_size, stride=stride, activation=int(bool(activation)))

    def _sky_config(self):
        return self._cfg

    def fused_forward(self, x):
        return self.forward(x)


class BottleneckBlock(NativeModule):
    """x + cv2(cv1(x)) when shortcut and in == out  (reference blocks.py:69-90)."""
    _sky_module = "BOTTLENECK"

    def __init__(self, in_channels, out_channels, shortcut=True, expansion=0.5):
        super().__init__()
        hidden_channels = int(out_channels * expansion)
        self.cv1 = ConvolutionBlock(in_channels, hidden_channels, 1, 1)
        self.cv2 = ConvolutionBlock(hidden_channels, out_channels, 3, 1)
        self.use_shortcut = shortcut and in_channels == out_channels
        self._cfg = dict(c_in=in_channels, c_out=out_channels, shortcut=int(bool(shortcut)), expansion=float(expansion))

    def _sky_config(self):
        return self._cfg


class CSPBlock(NativeModule):
    """cv3(cat(bottlenecks(cv1(x)), cv2(x)))  (reference blocks.py:93-123)."""
    _sky_module = "CSP"

    def __init__(self, in_channels, out_channels, num_blocks=1, shortcut=True, expansion=0.5):
        super().__init__()
        hidden_channels = int(out_channels * expansion)
        self.cv1 = ConvolutionBlock(in_channels, hidden_channels, 1, 1)
        self.cv2 = ConvolutionBlock(in_channels, hidden_channels, 1, 1)
        self.cv3 = ConvolutionBlock(2 * hidden_channels, out_channels, 1, 1)
        self.bottlenecks = nn.Sequential(*[BottleneckBlock(hidden_channels, hidden_channels, shortcut, 1.0)
                                           for _ in range(num_blocks)])
        self._cfg = dict(c_in=in_channels, c_out=out_channels, num_blocks=num_blocks, shortcut=int(bool(shortcut)),
                         expansion=float(expansion))

    def _sky_config(self):
        return self._cfg


class SPPBlock(NativeModule):
    """cv2(cat(y, mp5(y), mp9(y), mp13(y))), y = cv1(x)  (reference blocks.py:126-149)."""
    _sky_module = "SPP"

    def __init__(self, in_channels, out_channels, kernel_sizes=(5, 9, 13)):
        super().__init__()
        if tuple(kernel_sizes) != (5, 9, 13):
            raise NotImplementedError("the engine implements the reference's pooling pyramid (5, 9, 13) as a 5x5 cascade")
        hidden_channels = in_channels // 2
        self.cv1 = ConvolutionBlock(in_channels, hidden_channels, 1, 1)
        self.cv2 = ConvolutionBlock(hidden_channels * (len(kernel_sizes) + 1), out_channels, 1, 1)
        self.pooling = nn.ModuleList([Marker(f"MaxPool2d(k={k}, s=1, p={k // 2})") for k in kernel_sizes])
        self._cfg = dict(c_in=in_channels, c_out=out_channels)

    def _sky_config(self):
        return self._cfg


class FocusBlock(NativeModule):
    """space-to-depth (TL, BL, TR, BR) + ConvolutionBlock  (reference blocks.py:152-182)."""
    _sky_module = "FOCUS"

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, activation=True):
        super().__init__()
        if stride != 1 or not activation:
            raise NotImplementedError("FocusBlock is used with stride 1 and SiLU (backbone.py:48)")
        self.conv = ConvolutionBlock(in_channels * 4, out_channels, kernel_size, stride, padding, activation=activation)
        self._cfg = dict(c_in=in_channels, c_out=out_channels, kernel_size=kernel_size)

    def _sky_config(self):
        return self._cfg
