"""Positional encodings (aps/asr/transformer/pose.py): sinusoid tables keep the frozen `div_term`
parameter; the abs variant's add runs in aps_posenc_add."""
import math

import torch as th
import torch.nn as nn

from aps_amd import _native as nat
from aps_amd.libs import Register
from aps_amd.nn_ops import posenc_add

PosEncodings = Register("pos_encodings")


def get_xfmr_pose(pose: str, dim: int, **kwargs) -> nn.Module:
    if pose not in PosEncodings:
        raise ValueError(f"Unsupported pose layer: {pose}")
    return PosEncodings[pose](dim, **kwargs)


@PosEncodings.register("xl")
class SinPosEncoding(nn.Module):
    """sinusoid encodings, interleaved (sin, cos) (pose.py:27-62); as pose "xl" it only produces
    the 2T-1 x D table the XL attention projects (a <= 2T-1 row table: torch ops)"""

    def __init__(self, embed_dim: int, dropout: float = 0.0) -> None:
        super(SinPosEncoding, self).__init__()
        div_term = th.exp(-math.log(10000.0) * th.arange(0, embed_dim, 2.0) / embed_dim)
        self.div_term = nn.Parameter(div_term, requires_grad=False)
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, position: th.Tensor) -> th.Tensor:
        """T positions -> T x D"""
        sequence = position[:, None] * self.div_term
        table = th.stack([th.sin(sequence), th.cos(sequence)], dim=-1).view(position.shape[0], -1)
        return self._drop(table)

    def _drop(self, x: th.Tensor) -> th.Tensor:
        from aps_amd.grad_ops import dropout
        return dropout(x, self.dropout)

    def table(self, nframes: int) -> th.Tensor:
        """positions 0 .. 2T-2 (encoder.py:96-98)"""
        return self.forward(th.arange(0, 2 * nframes - 1, 1.0, device=self.div_term.device))


@PosEncodings.register("rel")
class RelPosEncoding(nn.Module):
    """learnt relative position embeddings, clamped to [-lradius, rradius] (pose.py:65-88)"""

    def __init__(self, embed_dim: int, dropout: float = 0.0, lradius: int = 128,
                 rradius: int = 128) -> None:
        super(RelPosEncoding, self).__init__()
        self.embed = nn.Embedding(lradius + rradius + 1, embed_dim)
        self.dropout = nn.Dropout(p=dropout)
        self.lradius, self.rradius = lradius, rradius

    def forward(self, position: th.Tensor) -> th.Tensor:
        """T (integer offsets) -> T x D; a row gather of a <= 2T-1 row table: torch indexing"""
        from aps_amd.grad_ops import GatherRowsFn, dropout
        position = th.clamp(position, max=self.rradius, min=-self.lradius)
        if nat.needs_grad(self.embed.weight):
            return dropout(GatherRowsFn.apply(self.embed.weight, position + self.lradius),
                           self.dropout)
        return dropout(self.embed.weight.detach()[position + self.lradius], self.dropout)

    def table(self, nframes: int) -> th.Tensor:
        """offsets -T+1 .. T-1 -> 2T-1 x D (what the encoder hands to every layer,
        encoder.py:91-95)"""
        if nframes - 1 <= min(self.lradius, self.rradius) and not nat.needs_grad(self.embed.weight) and \
                not (self.dropout.p > 0 and self.training):
            # no offset is clamped: the gathered rows ARE rows lradius - T + 1 .. lradius + T - 1 of the
            # embedding, in order -- a view, where the reference's arange / clamp / add / gather
            # (pose.py:78-88) is four launches of ~5 us inside every 3 ms step
            return self.embed.weight.detach()[self.lradius - nframes + 1:self.lradius + nframes]
        return self.forward(th.arange(-nframes + 1, nframes, device=self.embed.weight.device))


@PosEncodings.register("abs")
class InputSinPosEncoding(SinPosEncoding):
    """x * factor + sinusoid (pose.py:93-118)"""

    def __init__(self, embed_dim: int, dropout: float = 0.0, scaled: bool = False) -> None:
        super(InputSinPosEncoding, self).__init__(embed_dim, dropout=dropout)
        self.factor = embed_dim**0.5 if scaled else 1

    def add(self, inp: th.Tensor, t: int = 0) -> th.Tensor:
        """batch-major N x T x D -> N x T x D"""
        return self._drop(posenc_add(inp, self.div_term, float(self.factor), t))

    def forward(self, inp: th.Tensor, t: int = 0) -> th.Tensor:
        """N x T x D -> T x N x D (the reference's return layout)"""
        return self.add(inp, t).transpose(0, 1)
