"""
Transformer decoder forward (aps/asr/transformer/decoder.py:16-186) on the HIP kernels: token
embedding + sinusoid in one launch, causal self attention and cross attention on the attention
kernels, every projection / feed-forward on the fp32 MFMA GEMM with bias / activation / residual
(and, for pre-norm layers, the LayerNorm) in its epilogue.

Parameter names and shapes are the reference's (`vocab_embed.weight`, `decoder.layers.N.self_attn.
in_proj_weight` ... `decoder.norm.*`, `output.weight`): nn.MultiheadAttention / nn.LayerNorm /
nn.Linear modules hold them, the forward path does not call them.  Teacher-forced forward and
`step` (with `pre_emb`).  Under autograd / in train() mode every link has a HIP backward
(aps_amd/grad_ops.py): the projections, LayerNorms, the causal self attention (the context-window form
of the general attention adjoint), the attention over the encoder output (aps_attention_cross_backward),
the token embedding, and every dropout of the reference's layer -- on the attention weights, behind
the two attentions and inside / behind the feed-forward -- as counter-based masks.
"""
from typing import Dict, Optional, Tuple

import torch as th
import torch.nn as nn
from torch.nn import MultiheadAttention

from aps_amd.asr.transformer.impl import _check_activation, get_activation_fn
from aps_amd.grad_ops import ScaleAddFn, dropout, dropout_active
from aps_amd.asr.transformer.pose import get_xfmr_pose
from aps_amd.nn_ops import (attention_core, attention_cross, embedding_posenc, layernorm, linear,
                            posenc_add)


class TransformerDncoderLayer(nn.Module):
    """decoder layer: causal self attention, attention over the encoder output, feed-forward;
    pre- or post-norm (decoder.py:16-99; the class name is the reference's spelling)"""

    def __init__(self, att_dim: int, nhead: int, feedforward_dim: int = 2048,
                 pre_norm: bool = False, att_dropout: float = 0.1, ffn_dropout: float = 0.1,
                 activation: str = "relu") -> None:
        super(TransformerDncoderLayer, self).__init__()
        self.pre_norm = pre_norm
        self.activation = _check_activation(activation)
        self.self_attn = MultiheadAttention(att_dim, nhead, dropout=att_dropout)
        self.multihead_attn = MultiheadAttention(att_dim, nhead, dropout=att_dropout)
        self.feedforward = nn.Sequential(nn.Linear(att_dim, feedforward_dim),
                                         get_activation_fn(activation), nn.Dropout(ffn_dropout),
                                         nn.Linear(feedforward_dim, att_dim),
                                         nn.Dropout(ffn_dropout))
        self.norm1 = nn.LayerNorm(att_dim)
        self.norm2 = nn.LayerNorm(att_dim)
        self.norm3 = nn.LayerNorm(att_dim)
        self.dropout1 = nn.Dropout(ffn_dropout)
        self.dropout2 = nn.Dropout(ffn_dropout)
        self.nhead = nhead
        # nn.MultiheadAttention's `dropout` (on the attention weights) as modules that follow train() /
        # eval() (no parameters: the state dict is the reference's)
        self.self_attn_drop = nn.Dropout(att_dropout)
        self.cross_attn_drop = nn.Dropout(att_dropout)

    def memory_kv(self, memory: th.Tensor) -> th.Tensor:
        """key | value projections of the encoder output, N x S x D -> N x S x 2D"""
        att = self.multihead_attn
        D = att.embed_dim
        return linear(memory, att.in_proj_weight[D:], att.in_proj_bias[D:])

    def run(self, tgt: th.Tensor, memory: th.Tensor, tgt_len: Optional[th.Tensor],
            mem_len: Optional[th.Tensor], memory_mask: Optional[th.Tensor] = None) -> th.Tensor:
        """batch-major: tgt N x T x D, memory N x S x D -> N x T x D; memory_mask T x S additive"""
        sa, ca = self.self_attn, self.multihead_attn
        D = sa.embed_dim
        n1, n2, n3 = self.norm1, self.norm2, self.norm3
        pre = self.pre_norm

        def post(x, norm):
            return x if pre else layernorm(x, norm.weight, norm.bias, norm.eps)

        def out_proj(ctx, proj, residual, drop):
            """residual + dropout(out_proj(ctx)): one GEMM with the residual in its epilogue unless
            the dropout is active (train() mode)"""
            if dropout_active(drop):
                return ScaleAddFn.apply(dropout(linear(ctx, proj.weight, proj.bias), drop), residual, 1.0)
            return linear(ctx, proj.weight, proj.bias, residual=residual)

        # self attention under the sub-sequence mask (prep_sub_mask): key j <= query i
        qkv = linear(tgt, sa.in_proj_weight, sa.in_proj_bias, ln=n1 if pre else None)
        ctx = attention_core(qkv, self.nhead, tgt_len, chunk_size=1, lctx=-1, rctx=0,
                             dropout=self.self_attn_drop)
        tgt = post(out_proj(ctx, sa.out_proj, tgt, self.dropout1), n1)
        # attention over the encoder output
        q = linear(tgt, ca.in_proj_weight[:D], ca.in_proj_bias[:D], ln=n2 if pre else None)
        ctx = attention_cross(q, self.memory_kv(memory), self.nhead, mem_len, memory_mask,
                              dropout=self.cross_attn_drop)
        tgt = post(out_proj(ctx, ca.out_proj, tgt, self.dropout2), n2)
        # feed-forward: Linear - activation - Dropout - Linear - Dropout
        up, down = self.feedforward[0], self.feedforward[3]
        h = linear(tgt, up.weight, up.bias, act=self.activation, ln=n3 if pre else None)
        if dropout_active(self.feedforward[2], self.feedforward[4]):
            h = dropout(linear(dropout(h, self.feedforward[2]), down.weight, down.bias),
                        self.feedforward[4])
            return post(ScaleAddFn.apply(h, tgt, 1.0), n3)
        return post(linear(h, down.weight, down.bias, residual=tgt), n3)

    def forward(self, tgt: th.Tensor, memory: th.Tensor, tgt_mask: Optional[th.Tensor] = None,
                memory_mask: Optional[th.Tensor] = None,
                tgt_key_padding_mask: Optional[th.Tensor] = None,
                memory_key_padding_mask: Optional[th.Tensor] = None) -> th.Tensor:
        """reference call convention: T x N x D, S x N x D -> T x N x D.  The kernels build the
        sub-sequence mask themselves; padding masks must be length masks (as the reference's are).
        `memory_mask` (boolean or additive, T x S) is taken in eval and under autograd / in train() with
        dropout alike (round 5: the attention adjoints carry additive mask tensors -- data, no gradient
        into them; the reference's own decoder never passes one, aps/asr/transformer/decoder.py:133-186)"""
        if memory_mask is not None and memory_mask.dtype == th.bool:
            # nn.MultiheadAttention's boolean attn_mask (True = not visible) as the additive form
            memory_mask = th.zeros(memory_mask.shape, device=memory_mask.device).masked_fill_(
                memory_mask, float("-inf"))
        tl = None if tgt_key_padding_mask is None else (~tgt_key_padding_mask).sum(-1)
        ml = None if memory_key_padding_mask is None else (~memory_key_padding_mask).sum(-1)
        out = self.run(tgt.transpose(0, 1).contiguous(), memory.transpose(0, 1).contiguous(), tl, ml,
                       memory_mask)
        return out.transpose(0, 1)


class _DecoderStack(nn.Module):
    """parameter layout of torch's nn.TransformerDecoder: `layers.N.*` and optional `norm.*`"""

    def __init__(self, layer_kwargs: Dict, num_layers: int, norm: Optional[nn.Module]) -> None:
        super(_DecoderStack, self).__init__()
        self.layers = nn.ModuleList(
            [TransformerDncoderLayer(**layer_kwargs) for _ in range(num_layers)])
        self.norm = norm


class TorchTransformerDecoder(nn.Module):
    """Transformer decoder with absolute position encodings (decoder.py:102-186)"""

    def __init__(self, vocab_size: int, pose_kwargs: Dict = {}, arch_kwargs: Dict = {},
                 num_layers: int = 6) -> None:
        super(TorchTransformerDecoder, self).__init__()
        att_dim = arch_kwargs["att_dim"]
        self.vocab_embed = nn.Embedding(vocab_size, att_dim)
        self.abs_pos_enc = get_xfmr_pose("abs", att_dim, **pose_kwargs)
        final_norm = nn.LayerNorm(att_dim) if arch_kwargs.get("pre_norm") else None
        self.decoder = _DecoderStack(arch_kwargs, num_layers, final_norm)
        self.output = nn.Linear(att_dim, vocab_size, bias=False)
        self.vocab_size = vocab_size

    def step(self, enc_out: th.Tensor, tgt_pad: th.Tensor, enc_len: Optional[th.Tensor] = None,
             tgt_len: Optional[th.Tensor] = None, pre_emb: Optional[th.Tensor] = None,
             out_idx: Optional[int] = None) -> Tuple[th.Tensor, th.Tensor]:
        """enc_out T x N x D, tgt_pad N x To (, pre_emb T' x N x D) -> (dec_out T'+To x N x V or
        N x V for `out_idx`, tgt_emb T'+To x N x D)  (decoder.py:128-166)"""
        pose = self.abs_pos_enc
        offset = 0 if pre_emb is None else pre_emb.shape[0]
        emb = embedding_posenc(self.vocab_embed.weight, tgt_pad, pose.div_term, float(pose.factor),
                               offset)  # N x To x D
        emb = dropout(emb, pose.dropout)  # (InputSinPosEncoding's dropout, pose.py:93-118)
        if pre_emb is not None:
            emb = th.cat([pre_emb.transpose(0, 1), emb], 1)
        memory = enc_out.transpose(0, 1).contiguous()
        x = emb
        for layer in self.decoder.layers:
            x = layer.run(x, memory, tgt_len, enc_len)
        norm = self.decoder.norm
        if out_idx is not None:
            x = x[:, out_idx]
        if norm is not None:  # the final LayerNorm rides inside the output projection
            out = linear(x, self.output.weight, None, ln=norm)
        else:
            out = linear(x, self.output.weight, None)
        return (out if out_idx is not None else out.transpose(0, 1)), emb.transpose(0, 1)

    def forward(self, enc_out: th.Tensor, enc_len: Optional[th.Tensor], tgt_pad: th.Tensor,
                tgt_len: Optional[th.Tensor]) -> th.Tensor:
        """enc_out N x T x D, tgt_pad N x To -> N x To x V (decoder.py:168-186)"""
        dec_out, _ = self.step(enc_out.transpose(0, 1), tgt_pad, enc_len=enc_len, tgt_len=tgt_len)
        return dec_out.transpose(0, 1)
