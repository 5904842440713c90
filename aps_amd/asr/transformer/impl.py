"""
Multi-head self attention and transformer encoder layers -- the surface of
aps/asr/transformer/impl.py for the absolute-position transformer ("xfmr_abs"), on the kernels of
aps_amd/csrc/nn.hip.  Parameter names follow the reference (in_proj_weight / in_proj_bias /
out_proj, feedforward.0 / .3, norm1 / norm2, layers.N, norm) so checkpoints load.

Per layer: 4 GEMM launches (QKV projection; output projection + residual; FFN up + ReLU;
FFN down + residual), 1 attention-core launch and 2 LayerNorm launches; every elementwise op of
the reference (bias, ReLU, residual adds, scaling) is an epilogue of one of them.
Activations are kept batch-major (N x T x D) between layers; the reference's T x N x D layout is
accepted and returned at the module boundaries as transposed views.

Not built yet: relative / XL attention, conformer layers ("next").
"""
import copy
from typing import Dict, Optional

import torch as th
import torch.nn as nn

from aps_amd.libs import Register
from aps_amd.nn_ops import attention_core, layernorm, linear

TransformerEncoderLayers = Register("xfmr_encoder_layer")


def _eval_only(module: nn.Module, *dropouts: nn.Dropout) -> None:
    if module.training and any(d.p > 0 for d in dropouts):
        raise NotImplementedError("aps_amd encoder: forward (eval / dropout 0) path only")


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
               residual: Optional[th.Tensor] = None) -> th.Tensor:
        """self attention on batch-major x N x T x E; `residual` is added by the out-proj GEMM"""
        _eval_only(self, self.dropout)
        qkv = linear(x, self.in_proj_weight, self.in_proj_bias)
        ctx = attention_core(qkv, self.num_heads, lens)
        return linear(ctx, self.out_proj.weight, self.out_proj.bias, residual=residual)

    def forward(self, query, key, value, placehold=None, key_padding_mask=None, attn_mask=None):
        """L x N x E self attention (query is key is value) -> [context L x N x E]"""
        if key is not query or value is not query:
            raise NotImplementedError("aps_amd: self attention only (query = key = value)")
        if attn_mask is not None:
            raise NotImplementedError("aps_amd: additive attention masks (lctx/rctx) are not built")
        lens = None
        if key_padding_mask is not None:
            # padding masks of the encoder are length masks (padding_mask(inp_len))
            lens = (~key_padding_mask).sum(-1)
        out = self.attend(query.transpose(0, 1).contiguous(), lens)
        return [out.transpose(0, 1)]


class ApsTransformerEncoderLayer(nn.Module):
    """Post-/pre-norm transformer encoder layer (impl.py:377-429)"""

    def __init__(self, att_dim: int, self_attn: nn.Module, feedforward_dim: int = 2048,
                 dropout: float = 0.1, activation: str = "relu", pre_norm: bool = False) -> None:
        super(ApsTransformerEncoderLayer, self).__init__()
        if activation != "relu":
            raise NotImplementedError("aps_amd encoder: relu feed-forward only")
        self.self_attn = self_attn
        self.feedforward = nn.Sequential(nn.Linear(att_dim, feedforward_dim), nn.ReLU(),
                                         nn.Dropout(dropout), nn.Linear(feedforward_dim, att_dim),
                                         nn.Dropout(dropout))
        self.norm1 = nn.LayerNorm(att_dim)
        self.norm2 = nn.LayerNorm(att_dim)
        self.dropout = nn.Dropout(dropout)
        self.pre_norm = pre_norm

    def _ffn(self, x: th.Tensor, residual: th.Tensor) -> th.Tensor:
        up, down = self.feedforward[0], self.feedforward[3]
        h = linear(x, up.weight, up.bias, relu=True)
        return linear(h, down.weight, down.bias, residual=residual)

    def run(self, src: th.Tensor, lens: Optional[th.Tensor]) -> th.Tensor:
        """batch-major N x T x D -> N x T x D"""
        _eval_only(self, self.dropout, self.feedforward[2], self.feedforward[4])
        n1, n2 = self.norm1, self.norm2
        if self.pre_norm:
            inp = layernorm(src, n1.weight, n1.bias, n1.eps)
            src = self.self_attn.attend(inp, lens, residual=src)
            return self._ffn(layernorm(src, n2.weight, n2.bias, n2.eps), residual=src)
        src = self.self_attn.attend(src, lens, residual=src)     # src + att
        src = layernorm(src, n1.weight, n1.bias, n1.eps)
        return layernorm(self._ffn(src, residual=src), n2.weight, n2.bias, n2.eps)

    def forward(self, src, inj_pose=None, src_mask=None, src_key_padding_mask=None):
        """T x N x D -> T x N x D"""
        if src_mask is not None:
            raise NotImplementedError("aps_amd: additive attention masks (lctx/rctx) are not built")
        lens = None if src_key_padding_mask is None else (~src_key_padding_mask).sum(-1)
        return self.run(src.transpose(0, 1).contiguous(), lens).transpose(0, 1)


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


class ApsTransformerEncoder(nn.Module):
    """Stack of N encoder layers (+ final norm for pre-norm) (impl.py:718-756)"""

    def __init__(self, encoder_layer: nn.Module, num_layers: int,
                 norm: Optional[nn.Module] = None) -> None:
        super(ApsTransformerEncoder, self).__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = norm

    def run(self, x: th.Tensor, lens: Optional[th.Tensor]) -> th.Tensor:
        """batch-major N x T x D"""
        for mod in self.layers:
            x = mod.run(x, lens)
        if self.norm is not None:
            x = layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x

    def forward(self, src, inj_pose=None, src_mask=None, src_key_padding_mask=None):
        """T x N x D -> T x N x D"""
        if src_mask is not None:
            raise NotImplementedError("aps_amd: additive attention masks (lctx/rctx) are not built")
        lens = None if src_key_padding_mask is None else (~src_key_padding_mask).sum(-1)
        return self.run(src.transpose(0, 1).contiguous(), lens).transpose(0, 1)


def get_xfmr_encoder(arch: str, pose: str, num_layers: int, arch_kwargs: Dict) -> nn.Module:
    """factory (impl.py:759-787)"""
    name = f"{arch}_{pose}"
    if name not in TransformerEncoderLayers:
        raise ValueError(f"Unknown type of the encoders: {name}")
    att_dim = arch_kwargs["att_dim"]
    final_norm = nn.LayerNorm(att_dim) if arch_kwargs.get("pre_norm", False) else None
    return ApsTransformerEncoder(TransformerEncoderLayers[name](**arch_kwargs), num_layers,
                                 norm=final_norm)
