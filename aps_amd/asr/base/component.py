"""
Conv2d block (Conv2d -> Norm -> ReLU) and the 2-D normalisation wrapper of
aps/asr/base/component.py:117-142, 251-307 (parameter names `conv`, `norm.norm`).  The
convolution itself is a MIOpen library call through torch; this file is the
host plumbing and the output-length arithmetic, which is integer exact.
"""
from typing import Tuple, Union

import torch as th
import torch.nn as nn
import torch.nn.functional as tf


class Normalize2d(nn.Module):
    """BatchNorm2d / InstanceNorm2d wrapper"""

    def __init__(self, name: str, inp_features: int):
        super(Normalize2d, self).__init__()
        name = name.upper()
        if name not in ["BN", "IN"]:
            raise ValueError(f"Unknown type of Normalize2d: {name}")
        self.norm = nn.BatchNorm2d(inp_features) if name == "BN" else nn.InstanceNorm2d(inp_features)

    def __repr__(self) -> str:
        return str(self.norm)

    def forward(self, inp: th.Tensor) -> th.Tensor:
        return self.norm(inp)


class Conv2d(nn.Module):
    """... -> Conv2d -> Norm -> ReLU -> ..."""
    Conv2dParam = Union[int, Tuple[int, int]]

    def __init__(self,
                 in_channels: int,
                 out_channels: int,
                 kernel_size: Conv2dParam = 3,
                 stride: Conv2dParam = 2,
                 dilation: Conv2dParam = 1,
                 norm: str = "BN",
                 for_streaming: bool = False):
        super(Conv2d, self).__init__()

        def int2tuple(inp):
            return (inp, inp) if isinstance(inp, int) else inp

        kernel_size = int2tuple(kernel_size)
        dilation = int2tuple(dilation)
        padding = tuple((d * (k - 1)) // 2 for d, k in zip(dilation, kernel_size))
        if for_streaming:
            padding = (0, padding[-1])
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride,
                              padding=padding, dilation=dilation)
        self.norm = Normalize2d(norm, out_channels)
        self.kernel_size = kernel_size
        self.padding = padding
        self.dilation = dilation
        self.stride = int2tuple(stride)

    def compute_outp_dim(self, dim: th.Tensor, axis: int) -> th.Tensor:
        """output length along `axis`; NB dilation * kernel as in the reference (:290-297)"""
        return th.div(dim + 2 * self.padding[axis] - self.dilation[axis] * self.kernel_size[axis],
                      self.stride[axis], rounding_mode="trunc") + 1

    def forward(self, inp: th.Tensor) -> th.Tensor:
        """N x C x T x F -> N x C' x T' x F'"""
        out = self.norm(self.conv(inp[:, None] if inp.dim() == 3 else inp))
        return tf.relu(out)
