"""
SSEBase / MaskNonLinear (aps/sse/base.py:50-156): argument checks and the mask activation table.
The activations themselves are applied inside kernels (aps_dccrn_mask); `code()` maps the
reference's names onto the kernel enum.
"""
from typing import List, Optional

import torch as th
import torch.nn as nn

NONLINEAR_CODES = {"none": 0, "relu": 1, "tanh": 2, "softplus": 3, "sigmoid": 4}
_ENABLE = {
    "positive": ["relu", "softplus", "sigmoid", "softmax"],
    "positive_wo_softmax": ["relu", "softplus", "sigmoid"],
    "positive_wo_softplus": ["relu", "sigmoid", "softmax"],
    "all": ["none", "relu", "tanh", "softplus", "sigmoid", "softmax"],
    "all_wo_softmax": ["none", "relu", "tanh", "softplus", "sigmoid"],
    "bounded": ["sigmoid", "softmax"],
    "unbounded": ["relu", "softplus"],
    "common": ["relu", "sigmoid", "softmax"],
}


class SSEBase(nn.Module):
    """base class of the separation / enhancement models (sse/base.py:66-109)"""

    def __init__(self, transform: Optional[nn.Module], training_mode: str = "freq"):
        super(SSEBase, self).__init__()
        assert training_mode in ["freq", "time"]
        self.enh_transform = transform
        self.training_mode = training_mode

    def check_args(self, mix: th.Tensor, training: bool = True, valid_dim: List[int] = [2]) -> None:
        if mix.dim() not in valid_dim:
            supported_dim = "/".join([str(d) for d in valid_dim])
            raise RuntimeError(f"Expects {supported_dim}D tensor " +
                               f"({'training' if training else 'inference'}), " +
                               f"got {mix.dim()} instead")

    def infer(self, mix: th.Tensor, mode: str = "freq"):
        raise NotImplementedError()


class MaskNonLinear(nn.Module):
    """mask activation selector (sse/base.py:112-156)"""

    def __init__(self, non_linear: str, enable: str = "all", scale: float = 1,
                 vmax: Optional[float] = None, vmin: Optional[float] = None) -> None:
        super(MaskNonLinear, self).__init__()
        if enable not in _ENABLE:
            raise ValueError(f"Unsupported enable set: {enable}")
        if non_linear not in _ENABLE[enable]:
            raise ValueError(f"Unsupported nonlinear: {non_linear}")
        self.name = non_linear
        self.max, self.min, self.scale = vmax, vmin, scale

    def code(self) -> int:
        """kernel enum; scaled / clamped / softmax masks are not built into the kernels"""
        if self.name not in NONLINEAR_CODES or self.scale != 1 or self.max is not None or \
                self.min is not None:
            raise NotImplementedError(f"aps_amd: mask non-linearity {self.name} with scale / clamp")
        return NONLINEAR_CODES[self.name]
