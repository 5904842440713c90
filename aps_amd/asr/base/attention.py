"""
Attention modules of the RNN decoder (aps/asr/base/attention.py:18-531): `padding_mask`, the
"ctx" / "dot" / "loc" single-head attentions and their multi-head forms "mhctx" / "mhdot" / "mhloc"
with the reference's parameters (`enc_proj`, `key_proj`, `dec_proj`, `w`, `att`, `F`, `ctx_proj`).
A step is the decoder-state projection on the GEMM and one `aps_att_step` (`aps_att_step_heads`:
all heads in one launch) that scores every encoder frame, applies the masked softmax and forms the
context vector (+ `ctx_proj` on the GEMM for the multi-head forms).

Under autograd (round 5: `cmd/train_am.py` on an `att` recipe back-propagates through
aps/asr/base/decoder.py:165-218) the projections stay on `linear` (HIP forward and adjoint) and the step's
few small ops -- tanh / dot scores, the location convolutions, the masked softmax, the weighted sum -- run
on torch's own differentiable GPU ops (`_autograd_step`): a documented torch fall-through of the TRAINING
path only (no aps_att_step adjoint kernels were written); inference never takes it.
"""
from typing import Optional, Tuple

import torch as th
import torch.nn as nn


from aps_amd import _native as nat
from aps_amd.libs import Register
from aps_amd.nn_ops import linear

AsrAtt = Register("asr_att")


def padding_mask(vec: th.Tensor, device: th.device = None) -> th.Tensor:
    """lengths N -> N x max(len) boolean mask, True on padded positions (attention.py:18-36)"""
    frames = th.arange(int(vec.max()), device=vec.device)
    mask = frames[None, :] >= vec.reshape(-1, 1)
    return mask if device is None else mask.to(device)


def att_instance(att_type: str, enc_dim: int, dec_dim: int, **kwargs) -> nn.Module:
    if att_type not in AsrAtt:
        raise RuntimeError(f"Unknown attention type: {att_type}")
    return AsrAtt[att_type](enc_dim, dec_dim, **kwargs)


class Attention(nn.Module):
    """shared step logic; subclasses hold the parameters and name the kernel mode"""

    mode = 0

    def __init__(self) -> None:
        super(Attention, self).__init__()
        self.clear()

    def clear(self) -> None:
        self.enc_part = None

    def _step_args(self) -> dict:
        return {}

    def _dec_part(self, dec_prev: th.Tensor) -> th.Tensor:
        return linear(dec_prev, self.dec_proj.weight, self.dec_proj.bias)

    def forward(self, enc_pad: th.Tensor, enc_len: Optional[th.Tensor], dec_prev: th.Tensor,
                ali_prev: Optional[th.Tensor]) -> Tuple[th.Tensor, th.Tensor]:
        """enc_pad N x Ti x D_enc, dec_prev N x D_dec, ali_prev N x Ti | None ->
        (ali N x Ti, ctx N x D_enc)"""
        if nat.needs_grad(enc_pad, dec_prev, ali_prev, *self.parameters()):
            return self._autograd_step(enc_pad, enc_len, dec_prev, ali_prev)
        nat.require_device(enc_pad, enc_len, dec_prev, ali_prev)
        lib = nat.load()
        N, T, D = enc_pad.shape
        if self.enc_part is None:  # once per utterance batch (cleared by the model's forward)
            self.enc_part = linear(enc_pad, self.enc_proj.weight, self.enc_proj.bias)
            self._enc_pad = nat.f32c(enc_pad)
            self._enc_len = None if enc_len is None else \
                enc_len.to(device=enc_pad.device, dtype=th.int64).contiguous()
        dec_part = self._dec_part(dec_prev)
        A = dec_part.shape[-1]
        ali = th.empty(N, T, device=enc_pad.device, dtype=th.float32)
        ctx = th.empty(N, D, device=enc_pad.device, dtype=th.float32)
        x = self._step_args()
        w = getattr(self, "w", None)
        rc = lib.aps_att_step(nat.ptr(self.enc_part), nat.ptr(self._enc_pad), nat.ptr(dec_part),
                              nat.ptr(None if w is None else nat.f32c(w.weight)),
                              nat.ptr(self._enc_len),
                              nat.ptr(None if ali_prev is None or self.mode != 2 else
                                      nat.f32c(ali_prev)),
                              nat.ptr(x.get("filter")), nat.ptr(x.get("filter_bias")),
                              nat.ptr(x.get("att")), nat.ptr(ali), nat.ptr(ctx), N, T, A, D,
                              x.get("C", 0), x.get("L", 0), self.mode, float(x.get("scale", 1.0)),
                              nat.stream_of(enc_pad))
        nat.check(rc, "aps_att_step")
        return ali, ctx

    @staticmethod
    def _masked_softmax(score: th.Tensor, enc_len: Optional[th.Tensor]) -> th.Tensor:
        """softmax over the last (frame) axis with the frames past each utterance's length at -inf
        (attention.py:57-70); score N x T or N x H x T"""
        if enc_len is not None:
            T = score.shape[-1]
            pad = th.arange(T, device=score.device)[None, :] >= enc_len.to(score.device)[:, None]
            score = score.masked_fill(pad if score.dim() == 2 else pad[:, None], float("-inf"))
        return th.softmax(score, dim=-1)

    @staticmethod
    def _uniform_alignment(shape, enc_len: Optional[th.Tensor], device) -> th.Tensor:
        """the location-aware forms' first alignment: uniform over the valid frames (attention.py:123-130)"""
        ali = th.ones(*shape, device=device)
        if enc_len is None:
            return ali / shape[-1]
        T = shape[-1]
        pad = th.arange(T, device=device)[None, :] >= enc_len.to(device)[:, None]
        ali = ali.masked_fill(pad if len(shape) == 2 else pad[:, None], 0)
        return ali / enc_len.to(device).reshape(-1, *([1] * (len(shape) - 1)))

    def _autograd_step(self, enc_pad, enc_len, dec_prev, ali_prev):
        """the step under autograd (module docstring): same arithmetic as aps_att_step, on `linear` + torch ops"""
        N, T, _ = enc_pad.shape
        if self.enc_part is None:
            self.enc_part = linear(enc_pad, self.enc_proj.weight, self.enc_proj.bias)  # N x T x A
        dec_part = self._dec_part(dec_prev)  # N x A
        if self.mode == 1:
            score = (self.enc_part * dec_part[:, None]).sum(-1) * self._step_args().get("scale", 1.0)
        else:
            summed = self.enc_part + dec_part[:, None]
            if self.mode == 2:
                if ali_prev is None:
                    ali_prev = self._uniform_alignment((N, T), enc_len, enc_pad.device)
                att_part = self.att(self.F(ali_prev[:, None]))  # N x A x T
                summed = summed + att_part.transpose(1, 2)
            score = th.tanh(summed).matmul(self.w.weight.view(-1))
        ali = self._masked_softmax(score, enc_len)
        return ali, th.sum(ali[..., None] * enc_pad, 1)


@AsrAtt.register("ctx")
class CtxAttention(Attention):
    """additive attention, Bahdanau et al. (attention.py:158-206)"""

    mode = 0

    def __init__(self, enc_dim: int, dec_dim: int, att_dim: int = 512) -> None:
        super(CtxAttention, self).__init__()
        self.enc_proj = nn.Linear(enc_dim, att_dim)
        self.dec_proj = nn.Linear(dec_dim, att_dim, bias=False)
        self.w = nn.Linear(att_dim, 1, bias=False)


@AsrAtt.register("dot")
class DotAttention(Attention):
    """(scaled) dot attention, LAS (attention.py:209-259)"""

    mode = 1

    def __init__(self, enc_dim: int, dec_dim: int, att_dim: int = 512, scaled: bool = True) -> None:
        super(DotAttention, self).__init__()
        self.enc_proj = nn.Linear(enc_dim, att_dim)
        self.dec_proj = nn.Linear(dec_dim, att_dim)
        self.att_dim = att_dim
        self.scaled = scaled

    def _step_args(self) -> dict:
        return {"scale": self.att_dim**-0.5 if self.scaled else 1.0}


@AsrAtt.register("loc")
class LocAttention(Attention):
    """location aware attention, Chorowski et al. (attention.py:76-155)"""

    mode = 2

    def __init__(self, enc_dim: int, dec_dim: int, att_dim: int = 512, conv_channels: int = 10,
                 loc_context: int = 64) -> None:
        super(LocAttention, self).__init__()
        self.enc_proj = nn.Linear(enc_dim, att_dim)
        self.dec_proj = nn.Linear(dec_dim, att_dim, bias=False)
        self.att = nn.Conv1d(conv_channels, att_dim, 1, bias=False)
        self.F = nn.Conv1d(1, conv_channels, loc_context * 2 + 1, stride=1, padding=loc_context)
        self.w = nn.Linear(att_dim, 1, bias=False)
        self.conv_channels, self.loc_context = conv_channels, loc_context

    def _step_args(self) -> dict:
        return {"filter": nat.f32c(self.F.weight).view(self.conv_channels, -1),
                "filter_bias": None if self.F.bias is None else nat.f32c(self.F.bias),
                "att": nat.f32c(self.att.weight).view(-1, self.conv_channels),
                "C": self.conv_channels, "L": self.loc_context}


class MultiHeadAttention(Attention):
    """shared step of the multi-head forms (attention.py:266-531): keys and values are separate
    projections of the encoder output, head h owns columns h A .. (h + 1) A of both, of the query
    and of the grouped 1 x 1 convolutions; the concatenated head contexts go through `ctx_proj`"""

    def clear(self) -> None:
        self.enc_part = None
        self.key_part = None

    def forward(self, enc_pad: th.Tensor, enc_len: Optional[th.Tensor], dec_prev: th.Tensor,
                ali_prev: Optional[th.Tensor]) -> Tuple[th.Tensor, th.Tensor]:
        """enc_pad N x Ti x D_enc, dec_prev N x D_dec, ali_prev N x H x Ti | None ->
        (ali N x H x Ti, ctx N x D_enc)"""
        if nat.needs_grad(enc_pad, dec_prev, ali_prev, *self.parameters()):
            return self._autograd_step(enc_pad, enc_len, dec_prev, ali_prev)
        nat.require_device(enc_pad, enc_len, dec_prev, ali_prev)
        lib = nat.load()
        N, T, _ = enc_pad.shape
        H, A = self.att_head, self.att_dim
        if self.enc_part is None:  # once per utterance batch (cleared by the model's forward)
            self.enc_part = linear(enc_pad, self.enc_proj.weight, self.enc_proj.bias)  # values
            self.key_part = linear(enc_pad, self.key_proj.weight, self.key_proj.bias)
            self._enc_len = None if enc_len is None else \
                enc_len.to(device=enc_pad.device, dtype=th.int64).contiguous()
        dec_part = self._dec_part(dec_prev)  # N x H A
        ali = th.empty(N, H, T, device=enc_pad.device, dtype=th.float32)
        ctx = th.empty(N, H * A, device=enc_pad.device, dtype=th.float32)
        x = self._step_args()
        w = getattr(self, "w", None)
        rc = lib.aps_att_step_heads(nat.ptr(self.key_part), nat.ptr(self.enc_part),
                                    nat.ptr(dec_part),
                                    nat.ptr(None if w is None else nat.f32c(w.weight)),
                                    nat.ptr(self._enc_len),
                                    nat.ptr(None if ali_prev is None or self.mode != 2 else
                                            nat.f32c(ali_prev)),
                                    nat.ptr(x.get("filter")), nat.ptr(x.get("filter_bias")),
                                    nat.ptr(x.get("att")), nat.ptr(ali), nat.ptr(ctx), N, T, H, A,
                                    A, x.get("C", 0), x.get("L", 0), self.mode,
                                    float(x.get("scale", 1.0)), nat.stream_of(enc_pad))
        nat.check(rc, "aps_att_step_heads")
        return ali, linear(ctx, self.ctx_proj.weight, self.ctx_proj.bias)

    def _autograd_step(self, enc_pad, enc_len, dec_prev, ali_prev):
        """the multi-head step under autograd (attention.py:286-531): keys / values / query per head, the
        grouped 1 x 1 convolutions as per-head products, `ctx_proj` on `linear`"""
        N, T, _ = enc_pad.shape
        H, A = self.att_head, self.att_dim
        if self.enc_part is None:
            self.enc_part = linear(enc_pad, self.enc_proj.weight, self.enc_proj.bias)  # values N x T x H A
            self.key_part = linear(enc_pad, self.key_proj.weight, self.key_proj.bias)
        val = self.enc_part.view(N, T, H, A).transpose(1, 2)   # N x H x T x A
        key = self.key_part.view(N, T, H, A).transpose(1, 2)   # N x H x T x A
        dec = self._dec_part(dec_prev).view(N, H, A)
        if self.mode == 1:
            score = (key * dec[:, :, None]).sum(-1) * self._step_args().get("scale", 1.0)
        else:
            summed = key + dec[:, :, None]
            if self.mode == 2:
                if ali_prev is None:
                    ali_prev = self._uniform_alignment((N, H, T), enc_len, enc_pad.device)
                att_part = self.att(self.F(ali_prev))  # N x H A x T (grouped per head)
                summed = summed + att_part.view(N, H, A, T).transpose(2, 3)
            # w: Conv1d(H A -> H, 1, groups = H): head h's score = w[h] . tanh(.)
            score = (th.tanh(summed) * self.w.weight.view(1, H, 1, A)).sum(-1)
        ali = self._masked_softmax(score, enc_len)   # N x H x T
        ctx = th.sum(ali[..., None] * val, -2).reshape(N, H * A)
        return ali, linear(ctx, self.ctx_proj.weight, self.ctx_proj.bias)


@AsrAtt.register("mhctx")
class MHCtxAttention(MultiHeadAttention):
    """multi-head context attention (attention.py:265-344)"""

    mode = 0

    def __init__(self, enc_dim: int, dec_dim: int, att_dim: int = 512, att_head: int = 4) -> None:
        super(MHCtxAttention, self).__init__()
        self.enc_proj = nn.Linear(enc_dim, att_dim * att_head)
        self.key_proj = nn.Linear(enc_dim, att_dim * att_head, bias=False)
        self.dec_proj = nn.Linear(dec_dim, att_dim * att_head, bias=False)
        self.ctx_proj = nn.Linear(att_dim * att_head, enc_dim)
        self.w = nn.Conv1d(att_dim * att_head, att_head, 1, groups=att_head, bias=False)
        self.att_dim, self.att_head = att_dim, att_head


@AsrAtt.register("mhdot")
class MHDotAttention(MultiHeadAttention):
    """multi-head (scaled) dot attention (attention.py:347-422)"""

    mode = 1

    def __init__(self, enc_dim: int, dec_dim: int, att_dim: int = 512, att_head: int = 4,
                 scaled: bool = True) -> None:
        super(MHDotAttention, self).__init__()
        self.enc_proj = nn.Linear(enc_dim, att_dim * att_head, bias=False)
        self.key_proj = nn.Linear(enc_dim, att_dim * att_head, bias=False)
        self.dec_proj = nn.Linear(dec_dim, att_dim * att_head)
        self.ctx_proj = nn.Linear(att_dim * att_head, enc_dim)
        self.att_dim, self.att_head, self.scaled = att_dim, att_head, scaled

    def _step_args(self) -> dict:
        return {"scale": self.att_dim**-0.5 if self.scaled else 1.0}


@AsrAtt.register("mhloc")
class MHLocAttention(MultiHeadAttention):
    """multi-head location aware attention (attention.py:425-531)"""

    mode = 2

    def __init__(self, enc_dim: int, dec_dim: int, att_dim: int = 512, conv_channels: int = 10,
                 loc_context: int = 64, att_head: int = 4) -> None:
        super(MHLocAttention, self).__init__()
        self.enc_proj = nn.Linear(enc_dim, att_dim * att_head)
        self.key_proj = nn.Linear(enc_dim, att_dim * att_head, bias=False)
        self.dec_proj = nn.Linear(dec_dim, att_dim * att_head, bias=False)
        self.att = nn.Conv1d(conv_channels * att_head, att_dim * att_head, 1, groups=att_head,
                             bias=False)
        self.F = nn.Conv1d(att_head, conv_channels * att_head, loc_context * 2 + 1, stride=1,
                           groups=att_head, padding=loc_context)
        self.w = nn.Conv1d(att_dim * att_head, att_head, 1, groups=att_head, bias=False)
        self.ctx_proj = nn.Linear(att_dim * att_head, enc_dim)
        self.att_dim, self.att_head = att_dim, att_head
        self.conv_channels, self.loc_context = conv_channels, loc_context

    def _step_args(self) -> dict:
        HC = self.conv_channels * self.att_head
        return {"filter": nat.f32c(self.F.weight).view(HC, -1),
                "filter_bias": None if self.F.bias is None else nat.f32c(self.F.bias),
                "att": nat.f32c(self.att.weight).view(-1, self.conv_channels),
                "C": self.conv_channels, "L": self.loc_context}
