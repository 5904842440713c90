"""
CPU oracle for the transformer encoder forward (xfmr, absolute positions) -- TEST INFRASTRUCTURE.

Functional restatement, from a reference `state_dict`, of
    TransformerEncoder.forward           aps/asr/transformer/encoder.py:55-106
    Conv2dProj / Conv2dEncoder / Conv2d  aps/asr/transformer/proj.py:105-140,
                                         aps/asr/base/encoder.py:367-441, base/component.py:251-307
    InputSinPosEncoding                  aps/asr/transformer/pose.py:29-118
    ApsTransformerEncoderLayer (post/pre norm), ApsTransformerEncoder
                                         aps/asr/transformer/impl.py:377-429, 718-756
    multi-head self attention            impl.py:147-185 (delegates to
                                         torch.nn.functional.multi_head_attention_forward)
in eval mode (dropout off, BatchNorm running statistics).  Pinned by tests/golden/encoder_*.npz
recorded from the real reference.  Only tests / smoke / bench's cpu_baseline may import this.
"""
import math

import torch
import torch.nn.functional as F


def conv_out_len(length, kernel=3, stride=2, padding=1, dilation=1):
    """Conv2d.compute_outp_dim (component.py:290-297): NB dilation * kernel, not (kernel - 1)"""
    return torch.div(length + 2 * padding - dilation * kernel, stride, rounding_mode="trunc") + 1


def conv2d_proj(sd, x, x_len, prefix="proj.conv.", num_layers=2):
    """x N x T x F -> N x T' x D, lengths"""
    h = x[:, None]
    for i in range(num_layers):
        p = f"{prefix}enc_layers.{i}."
        h = F.conv2d(h, sd[p + "conv.weight"], sd[p + "conv.bias"], stride=2, padding=1)
        h = F.batch_norm(h, sd[p + "norm.norm.running_mean"], sd[p + "norm.norm.running_var"],
                         sd[p + "norm.norm.weight"], sd[p + "norm.norm.bias"], False, 0.0, 1e-5)
        h = torch.relu(h)
        if x_len is not None:
            x_len = conv_out_len(x_len)
    N, _, T, _ = h.shape
    h = h.transpose(1, 2).contiguous().view(N, T, -1)
    return F.linear(h, sd[prefix + "outp.weight"], sd[prefix + "outp.bias"]), x_len


def sin_pos_enc(T, D, div_term=None, start=0):
    """SinPosEncoding._get_sin_pos_enc (pose.py:42-50): interleaved (sin, cos)"""
    if div_term is None:
        div_term = torch.exp(-math.log(10000.0) * torch.arange(0, D, 2.0) / D)
    pos = torch.arange(start, start + T, 1.0)
    seq = pos[:, None] * div_term
    return torch.stack([torch.sin(seq), torch.cos(seq)], -1).view(T, -1)


def self_attention(sd, prefix, x, pad_mask, nhead, attn_mask=None):
    """x T x N x D; pad_mask N x T (True = padded); attn_mask additive T x T  ->  T x N x D"""
    T, N, D = x.shape
    dh = D // nhead
    qkv = F.linear(x, sd[prefix + "in_proj_weight"], sd[prefix + "in_proj_bias"])
    q, k, v = qkv.chunk(3, -1)
    q = q * (1.0 / math.sqrt(dh))
    # T x N x H x dh -> N x H x T x dh
    q, k, v = [m.reshape(T, N, nhead, dh).permute(1, 2, 0, 3) for m in (q, k, v)]
    score = torch.matmul(q, k.transpose(-1, -2))  # N x H x T x T
    if pad_mask is not None:
        score = score.masked_fill(pad_mask[:, None, None, :], float("-inf"))
    if attn_mask is not None:
        score = score + attn_mask[None, None]
    ctx = torch.matmul(torch.softmax(score, -1), v)  # N x H x T x dh
    ctx = ctx.permute(2, 0, 1, 3).reshape(T, N, D)
    return F.linear(ctx, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def context_mask(T, chunk_size=1, lctx=0, rctx=0):
    """prep_context_mask (transformer/utils.py:60-98): additive T x T mask, 0 / -inf"""
    if lctx < 0:
        lctx = T
    if rctx < 0:
        rctx = T
    index = torch.arange(T)
    seqs = index[None].repeat(T, 1)
    floor = torch.div(index, chunk_size, rounding_mode="floor")
    right = (floor + rctx + 1) * chunk_size
    left = torch.clamp_min((floor - lctx) * chunk_size, 0)
    hidden = (seqs >= right[:, None]) | (seqs < left[:, None])
    return torch.zeros(T, T).masked_fill(hidden, float("-inf"))


def xl_self_attention(sd, prefix, x, pad_mask, nhead, sin_pose, attn_mask=None):
    """XlMultiheadAttention.forward (impl.py:345-374): x T x N x D, sin_pose 2T-1 x D.
    NB the reference computes the logits from the VALUE projection (`dot_att(value, key, ...)`)."""
    T, N, D = x.shape
    dh = D // nhead
    qkv = F.linear(x, sd[prefix + "in_proj_weight"], sd[prefix + "in_proj_bias"])
    _, k, v = [m.reshape(T, N, nhead, dh) for m in qkv.chunk(3, -1)]
    term_ac = torch.einsum("lnhd,snhd->lnhs", v + sd[prefix + "rel_u"], k)
    rel_pos = F.linear(sin_pose, sd[prefix + "rel_proj.weight"]).view(-1, nhead, dh)
    term_bd = torch.einsum("lnhd,shd->lnhs", v + sd[prefix + "rel_v"], rel_pos)  # L N H 2L-1
    idx = torch.arange(T)[None, :] - torch.arange(T)[:, None] + T - 1  # digit_shift as a gather
    shifted = torch.gather(term_bd, -1, idx[:, None, None, :].expand(T, N, nhead, T))
    logit = (term_ac + shifted) / dh**0.5
    if pad_mask is not None:
        logit = logit.masked_fill(pad_mask[None, :, None, :], torch.finfo(torch.float32).min)
    if attn_mask is not None:
        logit = logit + attn_mask[:, None, None, :]
    ctx = torch.einsum("lnhs,snhd->lnhd", torch.softmax(logit, -1), v).reshape(T, N, D)
    return F.linear(ctx, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def any_attention(sd, prefix, x, pad_mask, nhead, rel=None, kind="abs", attn_mask=None):
    """abs | rel | xl self attention (kind follows the pose of the encoder)"""
    if kind == "xl":
        return xl_self_attention(sd, prefix, x, pad_mask, nhead, rel, attn_mask)
    if rel is not None:
        return rel_self_attention(sd, prefix, x, pad_mask, nhead, rel, attn_mask)
    return self_attention(sd, prefix, x, pad_mask, nhead, attn_mask)


def encoder_layer(sd, prefix, src, pad_mask, nhead, pre_norm=False, rel=None, kind=None,
                  attn_mask=None):
    """ApsTransformerEncoderLayer.forward (impl.py:402-429), relu feed-forward; rel: 2T-1 x dh
    table for the relative-position variant (xfmr_rel, impl.py:570-593)"""
    D = src.shape[-1]

    def ln(x, name):
        return F.layer_norm(x, (D,), sd[prefix + name + ".weight"], sd[prefix + name + ".bias"])

    def ffn(x):
        h = torch.relu(F.linear(x, sd[prefix + "feedforward.0.weight"],
                                sd[prefix + "feedforward.0.bias"]))
        return F.linear(h, sd[prefix + "feedforward.3.weight"], sd[prefix + "feedforward.3.bias"])

    inp = ln(src, "norm1") if pre_norm else src
    src = src + any_attention(sd, prefix + "self_attn.", inp, pad_mask, nhead, rel,
                              kind or ("abs" if rel is None else "rel"), attn_mask)
    if pre_norm:
        return src + ffn(ln(src, "norm2"))
    src = ln(src, "norm1")
    return ln(src + ffn(src), "norm2")


def xfmr_abs_encoder(sd, x, x_len, num_layers, nhead, pre_norm=False, scaled=False,
                     proj_layers=2):
    """TransformerEncoder("xfmr", proj="conv2d", pose="abs") forward: N x T x F -> N x T' x D"""
    h, h_len = conv2d_proj(sd, x, x_len, num_layers=proj_layers)
    N, T, D = h.shape
    pad_mask = None
    if h_len is not None:
        pad_mask = torch.arange(int(h_len.max().item()))[None] >= h_len[:, None]
    factor = D**0.5 if scaled else 1
    h = (h * factor + sin_pos_enc(T, D, sd.get("pose.div_term"))).transpose(0, 1)  # T x N x D
    for i in range(num_layers):
        h = encoder_layer(sd, f"encoder.layers.{i}.", h, pad_mask, nhead, pre_norm)
    if pre_norm:
        h = F.layer_norm(h, (D,), sd["encoder.norm.weight"], sd["encoder.norm.bias"])
    if "outp.weight" in sd:
        h = F.linear(h, sd["outp.weight"], sd["outp.bias"])
    return h.transpose(0, 1), h_len


# ----------------------------------------------------------------------------------------------
# conformer with relative position embeddings ("cfmr", pose "rel")
#   RelPosEncoding                    aps/asr/transformer/pose.py:65-88
#   RelMultiheadAttention             aps/asr/transformer/impl.py:225-296 (+ digit_shift,
#                                     aps/asr/transformer/utils.py:14-39, restated as the explicit
#                                     gather E[s - l + L - 1] it is equivalent to)
#   ApsConformerEncoderLayer          aps/asr/transformer/impl.py:432-541
# ----------------------------------------------------------------------------------------------
def rel_pos_table(sd, T, lradius, rradius):
    pos = torch.arange(-T + 1, T).clamp(min=-lradius, max=rradius) + lradius
    return sd["pose.embed.weight"][pos]  # 2T-1 x dh


def rel_self_attention(sd, prefix, x, pad_mask, nhead, rel, attn_mask=None):
    """x T x N x D, rel 2T-1 x dh -> T x N x D"""
    T, N, D = x.shape
    dh = D // nhead
    qkv = F.linear(x, sd[prefix + "in_proj_weight"], sd[prefix + "in_proj_bias"])
    q, k, v = [m.reshape(T, N, nhead, dh).permute(1, 2, 0, 3) for m in qkv.chunk(3, -1)]
    term_a = torch.matmul(q, k.transpose(-1, -2))  # N x H x T x T
    # relative term: row l uses E[s - l + T - 1]
    idx = torch.arange(T)[None, :] - torch.arange(T)[:, None] + T - 1  # T x T
    term_b = torch.einsum("nhld,lsd->nhls", q, rel[idx])
    score = (term_a + term_b) / dh**0.5
    if pad_mask is not None:
        score = score.masked_fill(pad_mask[:, None, None, :], torch.finfo(torch.float32).min)
    if attn_mask is not None:
        score = score + attn_mask[None, None]
    ctx = torch.matmul(torch.softmax(score, -1), v)
    ctx = ctx.permute(2, 0, 1, 3).reshape(T, N, D)
    return F.linear(ctx, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def conformer_layer(sd, p, src, pad_mask, nhead, rel, kernel_size=15, pre_norm=True, macaron=True,
                    kind=None, attn_mask=None, casual_conv1d=False):
    """conformer layer (impl.py:507-541), swish activations, eval mode; src T x N x D;
    rel None -> absolute-position attention (cfmr_abs)"""
    D = src.shape[-1]

    def ln(x, name):
        return F.layer_norm(x, (D,), sd[p + name + ".weight"], sd[p + name + ".bias"])

    def ffn(x, name):
        h = F.linear(x, sd[p + name + ".0.weight"], sd[p + name + ".0.bias"])
        h = h * torch.sigmoid(h)
        return F.linear(h, sd[p + name + ".3.weight"], sd[p + name + ".3.bias"])

    def conv(x):
        c = p + "convolution."
        h = x.permute(1, 2, 0)  # N x D x T
        if casual_conv1d:  # the module INPUT is padded on the left (impl.py:480, 499-500)
            h = F.pad(h, (kernel_size - 1, 0))
        h = F.conv1d(h, sd[c + "0.weight"], sd[c + "0.bias"])
        h = F.glu(h, dim=-2)
        h = F.conv1d(h, sd[c + "2.weight"], sd[c + "2.bias"],
                     padding=0 if casual_conv1d else (kernel_size - 1) // 2, groups=D)
        h = F.batch_norm(h, sd[c + "3.running_mean"], sd[c + "3.running_var"], sd[c + "3.weight"],
                         sd[c + "3.bias"], False, 0.0, 1e-5)
        h = h * torch.sigmoid(h)
        h = F.conv1d(h, sd[c + "5.weight"], sd[c + "5.bias"])
        return h.permute(2, 0, 1)

    def att(x):
        return any_attention(sd, p + "self_attn.", x, pad_mask, nhead, rel,
                             kind or ("abs" if rel is None else "rel"), attn_mask)

    factor = 0.5 if macaron else 1.0
    if pre_norm:
        if macaron:
            src = ffn(ln(src, "norm_ffn1"), "feedforward1") * factor + src
        src = src + att(ln(src, "norm_attn"))
        src = conv(ln(src, "norm_conv")) + src
        return ffn(ln(src, "norm_ffn2"), "feedforward2") * factor + src
    if macaron:
        src = ln(ffn(src, "feedforward1") * factor + src, "norm_ffn1")
    src = src + att(src)
    src = conv(ln(src, "norm_attn")) + src
    src = ln(src, "norm_conv")
    return ln(ffn(src, "feedforward2") * factor + src, "norm_ffn2")


def cfmr_rel_encoder(sd, x, x_len, num_layers, nhead, lradius, rradius, kernel_size=15,
                     proj_layers=2):
    """TransformerEncoder("cfmr", proj="conv2d", pose="rel") forward: N x T x F -> N x T' x D"""
    h, h_len = conv2d_proj(sd, x, x_len, num_layers=proj_layers)
    N, T, D = h.shape
    pad_mask = None
    if h_len is not None:
        pad_mask = torch.arange(int(h_len.max().item()))[None] >= h_len[:, None]
    rel = rel_pos_table(sd, T, lradius, rradius)
    h = h.transpose(0, 1)
    for i in range(num_layers):
        h = conformer_layer(sd, f"encoder.layers.{i}.", h, pad_mask, nhead, rel, kernel_size)
    if "encoder.norm.weight" in sd:
        h = F.layer_norm(h, (D,), sd["encoder.norm.weight"], sd["encoder.norm.bias"])
    if "outp.weight" in sd:
        h = F.linear(h, sd["outp.weight"], sd["outp.bias"])
    return h.transpose(0, 1), h_len


def linear_proj(sd, x, x_len, prefix="proj."):
    """LinearProj (proj.py:31-56): Linear -> GroupNorm(1, D) over the utterance -> ReLU"""
    h = F.linear(x, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])
    h = F.group_norm(h.transpose(1, 2), 1, sd[prefix + "norm.norm.weight"],
                     sd[prefix + "norm.norm.bias"], 1e-5).transpose(1, 2)
    return torch.relu(h), x_len


def conv1d_proj(sd, x, x_len, num_layers=2, kernel=3, stride=2, prefix="proj.conv."):
    """Conv1dProj / Conv1dEncoder / Conv1d blocks with BatchNorm1d (proj.py:59-101,
    encoder.py:310-364, component.py:192-248)"""
    h = x
    for i in range(num_layers):
        p = f"{prefix}enc_layers.{i}."
        pad = (kernel - 1) // 2
        h = F.conv1d(h.transpose(1, 2), sd[p + "conv.weight"], sd[p + "conv.bias"], stride, pad)
        h = F.batch_norm(h, sd[p + "norm.norm.running_mean"], sd[p + "norm.norm.running_var"],
                         sd[p + "norm.norm.weight"], sd[p + "norm.norm.bias"], False, 0.0, 1e-5)
        h = torch.relu(h).transpose(1, 2)
        if x_len is not None:
            x_len = torch.div(x_len + 2 * pad - (kernel - 1) - 1, stride, rounding_mode="trunc") + 1
    return h, x_len


def generic_encoder(sd, x, x_len, arch, pose, num_layers, nhead, lradius=128, rradius=128,
                    kernel_size=15, pre_norm=False, macaron=True, proj_layers=2, proj="conv2d",
                    window=None, casual_conv1d=False):
    """TransformerEncoder.forward (encoder.py:57-106) for arch xfmr | cfmr, pose abs | rel | xl,
    proj conv2d | linear | conv1d; window = (chunk_size, lctx, rctx) or None"""
    if proj == "conv2d":
        h, h_len = conv2d_proj(sd, x, x_len, num_layers=proj_layers)
    elif proj == "linear":
        h, h_len = linear_proj(sd, x, x_len)
    else:
        h, h_len = conv1d_proj(sd, x, x_len, num_layers=proj_layers)
    N, T, D = h.shape
    pad_mask = None
    if h_len is not None:
        pad_mask = torch.arange(int(h_len.max().item()))[None] >= h_len[:, None]
    rel = None
    if pose == "rel":
        rel = rel_pos_table(sd, T, lradius, rradius)
    elif pose == "xl":
        rel = sin_pos_enc(2 * T - 1, D, sd.get("pose.div_term"))  # positions 0 .. 2T-2
    else:
        h = h + sin_pos_enc(T, D, sd.get("pose.div_term"))
    attn_mask = None if window is None else context_mask(T, *window)
    h = h.transpose(0, 1)
    for i in range(num_layers):
        p = f"encoder.layers.{i}."
        if arch == "cfmr":
            h = conformer_layer(sd, p, h, pad_mask, nhead, rel, kernel_size, pre_norm, macaron,
                                pose, attn_mask, casual_conv1d)
        else:
            h = encoder_layer(sd, p, h, pad_mask, nhead, pre_norm, rel, pose, attn_mask)
    if "encoder.norm.weight" in sd:
        h = F.layer_norm(h, (D,), sd["encoder.norm.weight"], sd["encoder.norm.bias"])
    if "outp.weight" in sd:
        h = F.linear(h, sd["outp.weight"], sd["outp.bias"])
    return h.transpose(0, 1), h_len


# ------------------------------------------------------------------------------------------------
# Transformer decoder (aps/asr/transformer/decoder.py:16-186)
# ------------------------------------------------------------------------------------------------
def cross_attention(sd, prefix, tgt, memory, mem_pad_mask, nhead, attn_mask=None):
    """nn.MultiheadAttention(tgt, memory, memory) (decoder.py:78-86): tgt T x N x D, memory
    S x N x D, mem_pad_mask N x S (True = padded), attn_mask T x S (the layer's memory_mask:
    boolean, True = not visible, or additive float) -> T x N x D"""
    T, N, D = tgt.shape
    S = memory.shape[0]
    dh = D // nhead
    w, b = sd[prefix + "in_proj_weight"], sd[prefix + "in_proj_bias"]
    q = F.linear(tgt, w[:D], b[:D]) * (1.0 / math.sqrt(dh))
    k, v = F.linear(memory, w[D:], b[D:]).chunk(2, -1)
    q = q.reshape(T, N, nhead, dh).permute(1, 2, 0, 3)
    k, v = [m.reshape(S, N, nhead, dh).permute(1, 2, 0, 3) for m in (k, v)]
    score = torch.matmul(q, k.transpose(-1, -2))  # N x H x T x S
    if attn_mask is not None:
        if attn_mask.dtype == torch.bool:
            score = score.masked_fill(attn_mask[None, None], float("-inf"))
        else:
            score = score + attn_mask[None, None]
    if mem_pad_mask is not None:
        score = score.masked_fill(mem_pad_mask[:, None, None, :], float("-inf"))
    ctx = torch.matmul(torch.softmax(score, -1), v).permute(2, 0, 1, 3).reshape(T, N, D)
    return F.linear(ctx, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def decoder_layer(sd, p, tgt, memory, tgt_mask, tgt_pad_mask, mem_pad_mask, nhead, pre_norm,
                  memory_mask=None):
    """TransformerDncoderLayer.forward (decoder.py:46-99), eval mode"""

    def ln(x, name):
        return F.layer_norm(x, x.shape[-1:], sd[p + name + ".weight"], sd[p + name + ".bias"])

    skip = tgt
    x = ln(tgt, "norm1") if pre_norm else tgt
    x = skip + self_attention(sd, p + "self_attn.", x, tgt_pad_mask, nhead, tgt_mask)
    if not pre_norm:
        x = ln(x, "norm1")
    skip = x
    y = ln(x, "norm2") if pre_norm else x
    x = skip + cross_attention(sd, p + "multihead_attn.", y, memory, mem_pad_mask, nhead, memory_mask)
    if not pre_norm:
        x = ln(x, "norm2")
    skip = x
    y = ln(x, "norm3") if pre_norm else x
    h = torch.relu(F.linear(y, sd[p + "feedforward.0.weight"], sd[p + "feedforward.0.bias"]))
    x = skip + F.linear(h, sd[p + "feedforward.3.weight"], sd[p + "feedforward.3.bias"])
    return x if pre_norm else ln(x, "norm3")


def transformer_decoder(sd, enc_out, enc_len, tgt_pad, tgt_len, num_layers, nhead, pre_norm=False,
                        scaled=False, prefix=""):
    """TorchTransformerDecoder.forward (decoder.py:128-186): enc_out N x S x D, tgt_pad N x To
    -> N x To x V"""
    memory = enc_out.transpose(0, 1)
    To = tgt_pad.shape[1]
    D = sd[prefix + "vocab_embed.weight"].shape[1]
    emb = F.embedding(tgt_pad, sd[prefix + "vocab_embed.weight"])
    factor = D**0.5 if scaled else 1.0
    x = (emb * factor + sin_pos_enc(To, D, sd[prefix + "abs_pos_enc.div_term"])).transpose(0, 1)
    mem_mask = None if enc_len is None else \
        torch.arange(memory.shape[0])[None, :] >= enc_len[:, None]
    tgt_pad_mask = None if tgt_len is None else torch.arange(To)[None, :] >= tgt_len[:, None]
    sub = torch.triu(torch.ones(To, To), diagonal=1)
    sub = sub.masked_fill(sub == 1, float("-inf"))  # prep_sub_mask (transformer/utils.py:42-58)
    for i in range(num_layers):
        x = decoder_layer(sd, f"{prefix}decoder.layers.{i}.", x, memory, sub, tgt_pad_mask,
                          mem_mask, nhead, pre_norm)
    if pre_norm:
        x = F.layer_norm(x, x.shape[-1:], sd[prefix + "decoder.norm.weight"],
                         sd[prefix + "decoder.norm.bias"])
    return F.linear(x, sd[prefix + "output.weight"]).transpose(0, 1)
