"""
tf_masking / SSEBase / MaskNonLinear (aps/sse/base.py:23-156): the masking entry point of the
separation models, argument checks and the mask activation table.  DCCRN applies its activation
inside a kernel (aps_dccrn_mask; `code()` maps the reference's names onto the kernel enum).
"""
from typing import List, Optional

import torch as th
import torch.nn as nn
import torch.nn.functional as tf

from aps_amd import _native as nat
from aps_amd.ops import tf_mask_store
from aps_amd.spectrogram import packed_view, store_of


def tf_masking(mix_stft: th.Tensor, src_mask: th.Tensor, channel: int = 0) -> th.Tensor:
    """TF masking (sse/base.py:23-47): mix_stft N x (C) x F x T x 2, src_mask N x F x T (real) or
    N x F x T x 2 (complex) -> N x F x T x 2.  One launch (aps_tf_mask); differentiable w.r.t. the
    mask and the spectrogram (aps_tf_mask_backward)."""
    stft_dim, mask_dim = mix_stft.dim(), src_mask.dim()
    assert stft_dim in [4, 5]
    assert mask_dim in [3, 4]
    if stft_dim == 5:
        mix_stft = mix_stft[:, channel]
    if mask_dim == 4:
        assert src_mask.shape[-1] == 2
    return packed_view(tf_mask_store(store_of(mix_stft), src_mask))

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


def _mask_nonlinear(inp: th.Tensor, code: int, scale: float, vmin: float, vmax: float) -> th.Tensor:
    nat.require_device(inp.detach())
    x = nat.f32c(inp.detach())
    out = th.empty_like(x)
    rc = nat.load().aps_mask_nonlinear(nat.ptr(x), nat.ptr(out), x.shape[0], x.numel() // x.shape[0],
                                       code, scale, vmin, vmax, nat.stream_of(x))
    nat.check(rc, "aps_mask_nonlinear")
    return out


class _MaskNonLinearFn(th.autograd.Function):
    """MaskNonLinear under autograd: aps_mask_nonlinear forward, aps_mask_nonlinear_backward (the clamp's
    pass-through, the activation's derivative, the softmax's Jacobian over the sources)"""

    @staticmethod
    def forward(ctx, inp, code, scale, vmin, vmax):
        ctx.save_for_backward(inp)
        ctx.args = (code, scale, vmin, vmax)
        return _mask_nonlinear(inp, code, scale, vmin, vmax)

    @staticmethod
    def backward(ctx, g_out):
        (inp,) = ctx.saved_tensors
        code, scale, vmin, vmax = ctx.args
        x, g = nat.f32c(inp.detach()), nat.f32c(g_out)
        g_x = th.empty_like(x)
        rc = nat.load().aps_mask_nonlinear_backward(nat.ptr(x), nat.ptr(g), nat.ptr(g_x), x.shape[0],
                                                    x.numel() // x.shape[0], code, scale, vmin, vmax,
                                                    nat.stream_of(x))
        nat.check(rc, "aps_mask_nonlinear_backward")
        return g_x, None, None, None, None


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

    def forward(self, inp: th.Tensor) -> th.Tensor:
        """(S) x N x ... -> same shape (sse/base.py:141-156): clamp(f(inp) * scale, vmin, vmax) in
        one launch (aps_mask_nonlinear); softmax is over the leading source axis"""
        if inp.dim() not in [3, 4]:
            raise RuntimeError(f"MaskNonLinear expects 3/4D tensor, got {inp.dim()}")
        code = 5 if self.name == "softmax" else NONLINEAR_CODES[self.name]
        inf = float("inf")
        vmin = -inf if self.min is None else float(self.min)
        vmax = inf if self.max is None else float(self.max)
        if nat.needs_grad(inp):
            return _MaskNonLinearFn.apply(inp, code, float(self.scale), vmin, vmax)
        return _mask_nonlinear(inp, code, float(self.scale), vmin, vmax)

    def code(self) -> int:
        """kernel enum; scaled / clamped / softmax masks are not built into the kernels"""
        if self.name not in NONLINEAR_CODES or self.scale != 1 or self.max is not None or \
                self.min is not None:
            raise NotImplementedError(f"aps_amd: mask non-linearity {self.name} with scale / clamp")
        return NONLINEAR_CODES[self.name]
