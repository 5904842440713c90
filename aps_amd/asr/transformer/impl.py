"""
Multi-head self attention and transformer encoder layers -- the surface of
aps/asr/transformer/impl.py for the transformer / conformer with absolute or learnt relative
positions ("xfmr_abs", "xfmr_rel", "cfmr_abs", "cfmr_rel"), on the kernels of aps_amd/csrc/nn.hip.
Parameter names follow the reference (in_proj_weight / in_proj_bias / out_proj, feedforward.0 / .3,
norm1 / norm2, feedforward1/2, convolution.N, norm_ffn1/attn/conv/ffn2, layers.N, norm) so
checkpoints load.

Per layer: 4 GEMM launches (QKV projection; output projection + residual; FFN up + ReLU;
FFN down + residual), 1 attention-core launch and 2 LayerNorm launches; every elementwise op of
the reference (bias, ReLU, residual adds, scaling) is an epilogue of one of them.
Activations are kept batch-major (N x T x D) between layers; the reference's T x N x D layout is
accepted and returned at the module boundaries as transposed views.

Conformer layer (pre-norm): 8 GEMMs (2 x macaron FFN up/down, QKV, out-proj, 2 pointwise convs),
1 attention core, 1 GLU + depthwise conv + BatchNorm + Swish kernel = 10 launches; the 4 LayerNorms
are folded into the projections that consume them (aps_linear_layernorm), the 0.5 macaron scaling,
the Swish of the FFNs and all residual adds are GEMM epilogues.  Post-norm layers keep the
LayerNorm kernel (its output is also the residual stream).

Transformer-XL attention ("*_xl"), relative attention and context windows (lctx / rctx / chunk)
are variants of the one attention launch; a tensor mask goes in as an additive mask, the causal
(`casual_conv1d`) convolution module is the same GLU + depthwise kernel with all context on the left.
"""
import copy
from typing import Dict, Optional

import torch as th
import torch.nn as nn

from aps_amd import _native as nat
from aps_amd import mega
from aps_amd.grad_ops import ScaleAddFn, dropout, dropout_active
from aps_amd.libs import Register
from aps_amd.nn_ops import attention_core, glu_dwconv, layernorm, linear

TransformerEncoderLayers = Register("xfmr_encoder_layer")


def _window_kwargs(window) -> Dict:
    if window is None:
        return {}
    if isinstance(window, th.Tensor):  # an arbitrary additive T x T mask
        return {"add_mask": window}
    chunk, lctx, rctx = window
    return {"chunk_size": chunk, "lctx": lctx, "rctx": rctx}


def window_of_mask(src_mask: Optional[th.Tensor]):
    """The reference hands layers an additive T x T mask built by prep_context_mask; the kernels
    take the (chunk_size, lctx, rctx) triple instead (`ContextMask` carrier: every attention
    kernel, no T x T tensor read).  A plain tensor is any additive mask and goes to the streaming
    attention kernel as it is."""
    if src_mask is None:
        return None
    if isinstance(src_mask, ContextMask):
        return src_mask.window
    if isinstance(src_mask, th.Tensor) and src_mask.dim() == 2:
        return src_mask.float()
    raise RuntimeError(f"attention mask must be a T x T tensor or a ContextMask, got {type(src_mask)}")


class ContextMask(object):
    """(chunk_size, lctx, rctx) of prep_context_mask (transformer/utils.py:60-98)"""

    def __init__(self, chunk_size: int = 1, lctx: int = -1, rctx: int = -1) -> None:
        self.window = (int(chunk_size), int(lctx), int(rctx))


class ApsMultiheadAttention(nn.Module):
    """Multi-head attention with the reference's parameters (impl.py:22-222)"""

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0, bias: bool = True,
                 use_torch: bool = True) -> None:
        super(ApsMultiheadAttention, self).__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == self.embed_dim, "embed_dim must be divisible by num_heads"
        self.in_proj_weight = nn.Parameter(th.empty(3 * embed_dim, embed_dim))
        nn.init.xavier_uniform_(self.in_proj_weight)
        if bias:
            self.in_proj_bias = nn.Parameter(th.empty(3 * embed_dim))
            nn.init.constant_(self.in_proj_bias, 0)
        else:
            self.register_parameter("in_proj_bias", None)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self.dropout = nn.Dropout(p=dropout)
        self.use_torch = use_torch

    def attend(self, x: th.Tensor, lens: Optional[th.Tensor],
               residual: Optional[th.Tensor] = None, rel: Optional[th.Tensor] = None,
               window: Optional[tuple] = None, ln: Optional[nn.LayerNorm] = None,
               out_drop: Optional[nn.Dropout] = None) -> th.Tensor:
        """self attention on batch-major x N x T x E; `residual` is added by the out-proj GEMM;
        rel (2T-1 x dh | 2T-1 x E sinusoids) is only consumed by the relative / XL subclasses;
        window = (chunk_size, lctx, rctx) context limits or None; ln = the pre-norm LayerNorm of
        x, folded into the QKV projection"""
        qkv = linear(x, self.in_proj_weight, self.in_proj_bias, ln=ln)
        # train() mode: dropout on the attention weights inside the attention (impl.py:104)
        ctx = attention_core(qkv, self.num_heads, lens, **self._rel_kwargs(rel),
                             **_window_kwargs(window), **self._drop_kwargs())
        if dropout_active(out_drop):  # src + dropout(att): the GEMM epilogue cannot carry the residual
            out = dropout(linear(ctx, self.out_proj.weight, self.out_proj.bias), out_drop)
            return out if residual is None else ScaleAddFn.apply(out, residual, 1.0)
        # (pre-norm: the sum feeds the next projection through its folded LayerNorm)
        return linear(ctx, self.out_proj.weight, self.out_proj.bias, residual=residual, chain=ln is not None)

    def _drop_kwargs(self) -> Dict:
        return {"dropout": self.dropout} if dropout_active(self.dropout) else {}

    uses_rel = False

    def _rel_kwargs(self, rel: Optional[th.Tensor]) -> Dict:
        return {"rel": rel} if self.uses_rel else {}

    def forward(self, query, key, value, placehold=None, key_padding_mask=None, attn_mask=None):
        """L x N x E self attention (query is key is value) -> [context L x N x E]"""
        if key is not query or value is not query:
            raise NotImplementedError("aps_amd: self attention only (query = key = value)")
        lens = None
        if key_padding_mask is not None:
            # padding masks of the encoder are length masks (padding_mask(inp_len))
            lens = (~key_padding_mask).sum(-1)
        if self.uses_rel:
            if placehold is None or placehold.shape[0] != 2 * query.shape[0] - 1:
                raise RuntimeError("relative attention: positional table must have 2L-1 rows")
        out = self.attend(query.transpose(0, 1).contiguous(), lens, rel=placehold,
                          window=window_of_mask(attn_mask))
        return [out.transpose(0, 1), None]


class RelMultiheadAttention(ApsMultiheadAttention):
    """Self attention with learnt relative position keys (Shaw et al.; impl.py:225-296):
    logits = (q k^T + shift(q E^T)) / sqrt(dh); the shifted term is evaluated inside the attention
    kernel as q_i . E[j - i + L - 1] (aps_attention_core), no L x 2L-1 matrix is materialised."""

    uses_rel = True

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0, bias: bool = True) -> None:
        super(RelMultiheadAttention, self).__init__(embed_dim, num_heads, dropout=dropout,
                                                    bias=bias, use_torch=False)

    def attend(self, x, lens, residual=None, rel=None, window=None, ln=None, out_drop=None):
        if rel is None:
            raise RuntimeError("RelMultiheadAttention: relative position table missing")
        return super().attend(x, lens, residual=residual, rel=rel, window=window, ln=ln,
                              out_drop=out_drop)


def get_relative_uv(shape, init: str = "xavier", std: float = 0.02) -> nn.Parameter:
    """trainable biases of the XL attention (impl.py:708-716)"""
    if init not in ["xavier", "uniform"]:
        raise ValueError(f"Unknown init method: {init}")
    rel_mat = th.Tensor(*shape)
    if init == "xavier":
        nn.init.xavier_uniform_(rel_mat)
    else:
        nn.init.normal_(rel_mat, std=std)
    return nn.Parameter(rel_mat)


class XlMultiheadAttention(ApsMultiheadAttention):
    """Transformer-XL attention (impl.py:299-374): logits = (q + u) k^T + shift((q + v) R^T) with
    R = rel_proj(sinusoids) per head.  As in the reference's forward, the vector playing "q" is
    the VALUE projection (impl.py:366) -- reproduced so that its checkpoints give its outputs."""

    uses_rel = True

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0, bias: bool = True,
                 rel_u: Optional[nn.Parameter] = None, rel_v: Optional[nn.Parameter] = None) -> None:
        super(XlMultiheadAttention, self).__init__(embed_dim, num_heads, dropout=dropout,
                                                   bias=bias, use_torch=False)
        if rel_u is None or rel_v is None:
            self.rel_u = get_relative_uv((self.num_heads, self.head_dim))
            self.rel_v = get_relative_uv((self.num_heads, self.head_dim))
        else:
            self.rel_u = rel_u
            self.rel_v = rel_v
        self.rel_proj = nn.Linear(embed_dim, embed_dim, bias=False)

    def _rel_kwargs(self, rel: Optional[th.Tensor]) -> Dict:
        if rel is None:
            raise RuntimeError("XlMultiheadAttention: sinusoid table (2T-1 x E) missing")
        table = linear(rel, self.rel_proj.weight)  # 2T-1 x E -> per head H x 2T-1 x dh
        table = table.view(-1, self.num_heads, self.head_dim).transpose(0, 1).contiguous()
        return {"rel": table, "rel_u": self.rel_u, "rel_v": self.rel_v, "query_from_value": True}


class ApsTransformerEncoderLayer(nn.Module):
    """Post-/pre-norm transformer encoder layer (impl.py:377-429)"""

    def __init__(self, att_dim: int, self_attn: nn.Module, feedforward_dim: int = 2048,
                 dropout: float = 0.1, activation: str = "relu", pre_norm: bool = False) -> None:
        super(ApsTransformerEncoderLayer, self).__init__()
        self.activation = _check_activation(activation)
        self.self_attn = self_attn
        self.feedforward = nn.Sequential(nn.Linear(att_dim, feedforward_dim),
                                         get_activation_fn(activation),
                                         nn.Dropout(dropout), nn.Linear(feedforward_dim, att_dim),
                                         nn.Dropout(dropout))
        self.norm1 = nn.LayerNorm(att_dim)
        self.norm2 = nn.LayerNorm(att_dim)
        self.dropout = nn.Dropout(dropout)
        self.pre_norm = pre_norm

    def _ffn(self, x: th.Tensor, residual: th.Tensor, ln: Optional[nn.LayerNorm] = None) -> th.Tensor:
        up, down = self.feedforward[0], self.feedforward[3]
        h = linear(x, up.weight, up.bias, act=self.activation, ln=ln, chain=True)
        if dropout_active(self.feedforward[2], self.feedforward[4]):  # train(): Linear-act-Drop-Linear-Drop
            h = dropout(linear(dropout(h, self.feedforward[2]), down.weight, down.bias),
                        self.feedforward[4])
            return ScaleAddFn.apply(h, residual, 1.0)
        return linear(h, down.weight, down.bias, residual=residual, chain=ln is not None)

    def run(self, src: th.Tensor, lens: Optional[th.Tensor], rel: Optional[th.Tensor] = None,
            window: Optional[tuple] = None) -> th.Tensor:
        """batch-major N x T x D -> N x T x D"""
        n1, n2 = self.norm1, self.norm2
        if self.pre_norm:  # both LayerNorms ride inside the projections that consume them
            src = self.self_attn.attend(src, lens, residual=src, rel=rel, window=window, ln=n1,
                                        out_drop=self.dropout)
            return self._ffn(src, residual=src, ln=n2)
        src = self.self_attn.attend(src, lens, residual=src, rel=rel, window=window,
                                    out_drop=self.dropout)  # src + dropout(att)
        src = layernorm(src, n1.weight, n1.bias, n1.eps)
        return layernorm(self._ffn(src, residual=src), n2.weight, n2.bias, n2.eps)

    def forward(self, src, inj_pose=None, src_mask=None, src_key_padding_mask=None):
        """T x N x D -> T x N x D"""
        lens = None if src_key_padding_mask is None else (~src_key_padding_mask).sum(-1)
        return self.run(src.transpose(0, 1).contiguous(), lens, rel=inj_pose,
                        window=window_of_mask(src_mask)).transpose(0, 1)


class Swish(nn.Module):
    """x * sigmoid(x) (impl.py:299-307); only a marker here, the GEMM / conv epilogues apply it"""

    def forward(self, inp: th.Tensor) -> th.Tensor:
        raise NotImplementedError("aps_amd: Swish is applied as a kernel epilogue")


def _check_activation(name: str) -> str:
    if name not in ("relu", "gelu", "swish"):
        raise RuntimeError(f"activation should be relu/gelu/swish, not {name}")
    return name


def get_activation_fn(activation: str) -> nn.Module:
    """activation marker modules (parameter free, keep the Sequential indices of impl.py:310-322)"""
    return {"relu": nn.ReLU, "gelu": nn.GELU, "swish": Swish}[_check_activation(activation)]()


class ApsConformerEncoderLayer(nn.Module):
    """Conformer encoder layer (impl.py:432-541): macaron FFN, MHSA, convolution module, FFN"""

    def __init__(self, att_dim: int, self_attn: nn.Module, feedforward_dim: int = 2048,
                 dropout: float = 0.1, kernel_size: int = 15, macaron: float = True,
                 pre_norm: bool = True, casual_conv1d: bool = False,
                 activation: str = "swish") -> None:
        super(ApsConformerEncoderLayer, self).__init__()
        assert kernel_size % 2 == 1
        self.activation = _check_activation(activation)
        self.self_attn = self_attn

        def ffn():
            return nn.Sequential(nn.Linear(att_dim, feedforward_dim),
                                 get_activation_fn(activation), nn.Dropout(dropout),
                                 nn.Linear(feedforward_dim, att_dim), nn.Dropout(dropout))

        if macaron:
            self.norm_ffn1 = nn.LayerNorm(att_dim)
            self.macaron_factor = 0.5
            self.feedforward1 = ffn()
        else:
            self.macaron_factor = 1
            self.norm_ffn1 = None
            self.feedforward1 = None
        self.convolution = nn.Sequential(
            nn.Conv1d(att_dim, att_dim * 2, 1), nn.GLU(dim=-2),
            nn.Conv1d(att_dim, att_dim, kernel_size, groups=att_dim,
                      padding=0 if casual_conv1d else (kernel_size - 1) // 2),
            nn.BatchNorm1d(att_dim),
            get_activation_fn(activation), nn.Conv1d(att_dim, att_dim, 1), nn.Dropout(p=dropout))
        self.norm_ffn2 = nn.LayerNorm(att_dim)
        self.feedforward2 = ffn()
        self.norm_attn = nn.LayerNorm(att_dim)
        self.norm_conv = nn.LayerNorm(att_dim)
        self.dropout = nn.Dropout(dropout)
        self.padding = kernel_size - 1 if casual_conv1d else 0  # left context of the causal form
        self.pre_norm = pre_norm
        self._bn_cache = None

    def _bn_affine(self):
        """eval-mode BatchNorm1d as (scale, shift), refreshed when its tensors change"""
        bn = self.convolution[3]
        parts = [bn.running_mean, bn.running_var, bn.weight, bn.bias]
        key = tuple((t.data_ptr(), t._version) for t in parts if t is not None)
        if self._bn_cache is None or self._bn_cache[0] != key:
            if bn.running_mean is None:
                raise NotImplementedError("aps_amd conformer: BatchNorm1d needs running statistics")
            scale = th.rsqrt(bn.running_var.detach().float() + bn.eps)
            if bn.weight is not None:
                scale = scale * bn.weight.detach().float()
            shift = -bn.running_mean.detach().float() * scale
            if bn.bias is not None:
                shift = shift + bn.bias.detach().float()
            self._bn_cache = (key, scale.contiguous(), shift.contiguous())
        return self._bn_cache[1], self._bn_cache[2]

    def _ffn(self, ffn: nn.Sequential, x: th.Tensor, residual: th.Tensor,
             ln: Optional[nn.LayerNorm] = None) -> th.Tensor:
        h = linear(x, ffn[0].weight, ffn[0].bias, act=self.activation, ln=ln, chain=True)
        if dropout_active(ffn[2], ffn[4]):  # train(): Linear-act-Drop-Linear-Drop, then * factor + src
            h = dropout(linear(dropout(h, ffn[2]), ffn[3].weight, ffn[3].bias), ffn[4])
            return ScaleAddFn.apply(h, residual, self.macaron_factor)
        return linear(h, ffn[3].weight, ffn[3].bias, alpha=self.macaron_factor, residual=residual,
                      chain=ln is not None)

    def conv_run(self, x: th.Tensor, residual: th.Tensor,
                 ln: Optional[nn.LayerNorm] = None) -> th.Tensor:
        """convolution module on batch-major N x T x D (+ residual in the last GEMM); ln = the
        LayerNorm in front of it, folded into the first pointwise projection"""
        c = self.convolution
        D = x.shape[-1]
        ln_params = () if ln is None else (ln.weight, ln.bias)
        if c[3].training or nat.needs_grad(x, residual, *ln_params, *c.parameters()):
            # training / autograd: the un-fused chain, every link with a HIP backward (grad_ops):
            # GEMM -> GLU + depthwise conv -> BatchNorm (batch statistics in train()) -> activation
            # -> GEMM
            from aps_amd.grad_ops import activation, batchnorm_rows
            h = linear(x, c[0].weight.view(2 * D, D), c[0].bias, ln=ln)
            h = glu_dwconv(h, c[2].weight, c[2].bias, None, None, act="none",
                           causal=self.padding > 0, pad_bias=c[0].bias)
            h = activation(batchnorm_rows(h, c[3]), self.activation)
            if dropout_active(c[6]):
                h = dropout(linear(h, c[5].weight.view(D, D), c[5].bias), c[6])
                return h if residual is None else ScaleAddFn.apply(h, residual, 1.0)
            return linear(h, c[5].weight.view(D, D), c[5].bias, residual=residual)
        h = linear(x, c[0].weight.view(2 * D, D), c[0].bias, ln=ln)
        scale, shift = self._bn_affine()
        h = glu_dwconv(h, c[2].weight, c[2].bias, scale, shift, act=self.activation,
                       causal=self.padding > 0, pad_bias=c[0].bias)
        return linear(h, c[5].weight.view(D, D), c[5].bias, residual=residual, chain=ln is not None)

    def conv(self, inp: th.Tensor) -> th.Tensor:
        """T x N x D -> T x N x D (impl.py:491-505)"""
        return self.conv_run(inp.transpose(0, 1).contiguous(), None).transpose(0, 1)

    def run(self, src: th.Tensor, lens: Optional[th.Tensor], rel: Optional[th.Tensor] = None,
            window: Optional[tuple] = None) -> th.Tensor:
        """batch-major N x T x D -> N x T x D"""

        def ln(m, x):
            return layernorm(x, m.weight, m.bias, m.eps)

        if self.pre_norm:  # every LayerNorm rides inside the projection that consumes it
            if self.feedforward1 is not None:
                src = self._ffn(self.feedforward1, src, src, ln=self.norm_ffn1)
            src = self.self_attn.attend(src, lens, residual=src, rel=rel, window=window,
                                        ln=self.norm_attn, out_drop=self.dropout)
            src = self.conv_run(src, src, ln=self.norm_conv)
            return self._ffn(self.feedforward2, src, src, ln=self.norm_ffn2)
        if self.feedforward1 is not None:
            src = ln(self.norm_ffn1, self._ffn(self.feedforward1, src, src))
        src = self.self_attn.attend(src, lens, residual=src, rel=rel, window=window,
                                    out_drop=self.dropout)
        src = self.conv_run(ln(self.norm_attn, src), src)
        src = ln(self.norm_conv, src)
        return ln(self.norm_ffn2, self._ffn(self.feedforward2, src, src))

    def forward(self, src, inj_pose=None, src_mask=None, src_key_padding_mask=None):
        """T x N x D -> T x N x D"""
        lens = None if src_key_padding_mask is None else (~src_key_padding_mask).sum(-1)
        return self.run(src.transpose(0, 1).contiguous(), lens, rel=inj_pose,
                        window=window_of_mask(src_mask)).transpose(0, 1)


@TransformerEncoderLayers.register("xfmr_abs")
class TransformerEncoderLayer(ApsTransformerEncoderLayer):
    """Standard transformer encoder layer with absolute positions (impl.py:544-568)"""

    def __init__(self, att_dim: int, nhead: int, feedforward_dim: int = 2048,
                 pre_norm: bool = False, att_dropout: float = 0.1, ffn_dropout: float = 0.1,
                 activation: str = "relu") -> None:
        self_attn = ApsMultiheadAttention(att_dim, nhead, dropout=att_dropout, use_torch=True)
        super(TransformerEncoderLayer, self).__init__(att_dim, self_attn,
                                                      feedforward_dim=feedforward_dim,
                                                      dropout=ffn_dropout, activation=activation,
                                                      pre_norm=pre_norm)


@TransformerEncoderLayers.register("xfmr_rel")
class TransformerRelEncoderLayer(ApsTransformerEncoderLayer):
    """Transformer encoder layer with learnt relative positions (impl.py:570-593)"""

    def __init__(self, att_dim: int, nhead: int, feedforward_dim: int = 2048,
                 att_dropout: float = 0.1, ffn_dropout: float = 0.1, activation: str = "relu",
                 pre_norm: bool = False) -> None:
        self_attn = RelMultiheadAttention(att_dim, nhead, dropout=att_dropout)
        super(TransformerRelEncoderLayer, self).__init__(att_dim, self_attn,
                                                         feedforward_dim=feedforward_dim,
                                                         dropout=ffn_dropout,
                                                         activation=activation, pre_norm=pre_norm)


@TransformerEncoderLayers.register("cfmr_abs")
class ConformerEncoderLayer(ApsConformerEncoderLayer):
    """Conformer encoder layer with absolute positions (impl.py:625-653)"""

    def __init__(self, att_dim: int, nhead: int, feedforward_dim: int = 2048,
                 att_dropout: float = 0.1, ffn_dropout: float = 0.1, kernel_size: int = 15,
                 macaron: bool = True, pre_norm: bool = True, activation: str = "swish") -> None:
        self_attn = ApsMultiheadAttention(att_dim, nhead, dropout=att_dropout, use_torch=True)
        super(ConformerEncoderLayer, self).__init__(att_dim, self_attn,
                                                    feedforward_dim=feedforward_dim,
                                                    dropout=ffn_dropout, activation=activation,
                                                    kernel_size=kernel_size, macaron=macaron,
                                                    pre_norm=pre_norm)


@TransformerEncoderLayers.register("cfmr_rel")
class ConformerRelEncoderLayer(ApsConformerEncoderLayer):
    """Conformer encoder layer with learnt relative positions (impl.py:656-681)"""

    def __init__(self, att_dim: int, nhead: int, feedforward_dim: int = 2048,
                 att_dropout: float = 0.1, ffn_dropout: float = 0.1, kernel_size: int = 15,
                 macaron: bool = True, pre_norm: bool = True, activation: str = "swish") -> None:
        self_attn = RelMultiheadAttention(att_dim, nhead, dropout=att_dropout)
        super(ConformerRelEncoderLayer, self).__init__(att_dim, self_attn,
                                                       feedforward_dim=feedforward_dim,
                                                       dropout=ffn_dropout, activation=activation,
                                                       kernel_size=kernel_size, macaron=macaron,
                                                       pre_norm=pre_norm)


@TransformerEncoderLayers.register("xfmr_xl")
class TransformerXLEncoderLayer(ApsTransformerEncoderLayer):
    """Transformer encoder layer with Transformer-XL attention (impl.py:596-622)"""

    def __init__(self, att_dim: int, nhead: int, feedforward_dim: int = 2048,
                 att_dropout: float = 0.1, ffn_dropout: float = 0.1, activation: str = "relu",
                 pre_norm: bool = False, rel_u: Optional[nn.Parameter] = None,
                 rel_v: Optional[nn.Parameter] = None) -> None:
        self_attn = XlMultiheadAttention(att_dim, nhead, dropout=att_dropout, rel_u=rel_u,
                                         rel_v=rel_v)
        super(TransformerXLEncoderLayer, self).__init__(att_dim, self_attn,
                                                        feedforward_dim=feedforward_dim,
                                                        dropout=ffn_dropout, activation=activation,
                                                        pre_norm=pre_norm)


@TransformerEncoderLayers.register("cfmr_xl")
class ConformerXLEncoderLayer(ApsConformerEncoderLayer):
    """Conformer encoder layer with Transformer-XL attention (impl.py:684-715)"""

    def __init__(self, att_dim: int, nhead: int, feedforward_dim: int = 2048,
                 att_dropout: float = 0.1, ffn_dropout: float = 0.1, kernel_size: int = 15,
                 macaron: bool = True, pre_norm: bool = True, activation: str = "swish",
                 rel_u: Optional[nn.Parameter] = None, rel_v: Optional[nn.Parameter] = None) -> None:
        self_attn = XlMultiheadAttention(att_dim, nhead, dropout=att_dropout, rel_u=rel_u,
                                         rel_v=rel_v)
        super(ConformerXLEncoderLayer, self).__init__(att_dim, self_attn,
                                                      feedforward_dim=feedforward_dim,
                                                      dropout=ffn_dropout, activation=activation,
                                                      kernel_size=kernel_size, macaron=macaron,
                                                      pre_norm=pre_norm)


class ApsTransformerEncoder(nn.Module):
    """Stack of N encoder layers (+ final norm for pre-norm) (impl.py:718-756)"""

    def __init__(self, encoder_layer: nn.Module, num_layers: int,
                 norm: Optional[nn.Module] = None) -> None:
        super(ApsTransformerEncoder, self).__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = norm

    def run(self, x: th.Tensor, lens: Optional[th.Tensor], rel: Optional[th.Tensor] = None,
            window: Optional[tuple] = None) -> th.Tensor:
        """batch-major N x T x D; rel = relative position table (2T-1 x dh) for "*_rel" layers,
        sinusoid table (2T-1 x D) for "*_xl" layers; window = (chunk_size, lctx, rctx) or None"""
        # the stack as ONE launch per batch, a workgroup per utterance (csrc/conformer_mega.hip), for the stacks and
        # shapes that kernel is built for, while several streams are launching (mega.wanted); otherwise -- and always
        # under autograd / in train() -- one launch per projection
        done = False
        if mega.wanted() and not nat.needs_grad(x, *self.parameters()) and mega.supported(self, x, rel, window):
            y = mega.conformer_stack(self, x, lens, rel)
            if y is not None:
                x, done = y, True
        if not done:
            for mod in self.layers:
                x = mod.run(x, lens, rel=rel, window=window)
        if self.norm is not None:
            x = layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x

    def forward(self, src, inj_pose=None, src_mask=None, src_key_padding_mask=None):
        """T x N x D -> T x N x D"""
        lens = None if src_key_padding_mask is None else (~src_key_padding_mask).sum(-1)
        return self.run(src.transpose(0, 1).contiguous(), lens, rel=inj_pose,
                        window=window_of_mask(src_mask)).transpose(0, 1)


def get_xfmr_encoder(arch: str, pose: str, num_layers: int, arch_kwargs: Dict) -> nn.Module:
    """factory (impl.py:759-787)"""
    name = f"{arch}_{pose}"
    if name not in TransformerEncoderLayers:
        raise ValueError(f"Unknown type of the encoders: {name}")
    att_dim = arch_kwargs["att_dim"]
    # as in the reference the final norm follows the *explicit* pre_norm kwarg only: a conformer
    # left at its pre_norm=True default gets none
    final_norm = nn.LayerNorm(att_dim) if arch_kwargs.get("pre_norm", False) else None
    if pose == "xl":  # optional tying of the XL biases across layers (impl.py:772-783)
        rel_u, rel_v = None, None
        if arch_kwargs.pop("tie", False):
            nhead = arch_kwargs["nhead"]
            rel_u = get_relative_uv((nhead, att_dim // nhead))
            rel_v = get_relative_uv((nhead, att_dim // nhead))
        arch_kwargs["rel_u"] = rel_u
        arch_kwargs["rel_v"] = rel_v
    return ApsTransformerEncoder(TransformerEncoderLayers[name](**arch_kwargs), num_layers,
                                 norm=final_norm)
