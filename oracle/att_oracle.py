"""
TEST INFRASTRUCTURE -- CPU restatement of the reference's RNN attention decoder forward
(aps/asr/base/decoder.py:69-218 with the attentions of aps/asr/base/attention.py:76-531) as
functional torch-CPU ops on a state_dict.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  Pinned by tests/test_oracle_encoder.py against the fixtures
att_decoder_* recorded from the reference's modules.
"""
import torch
import torch.nn.functional as F


def attention_step(sd, p, kind, enc_pad, enc_part, pad_mask, enc_len, dec_prev, ali_prev,
                   scaled=True, loc_context=0):
    """one attention call -> (ali N x T, ctx N x D)"""
    N, T, _ = enc_pad.shape
    dec_part = F.linear(dec_prev, sd[p + "dec_proj.weight"], sd.get(p + "dec_proj.bias"))
    if kind == "dot":  # attention.py:247-250
        score = torch.bmm(enc_part, dec_part[..., None]).squeeze(-1)
        if scaled:
            score = score / (enc_part.shape[-1]**0.5)
    else:
        s = enc_part + dec_part[:, None]
        if kind == "loc":  # attention.py:119-141
            if ali_prev is None:
                ali_prev = torch.ones(N, T)
                if enc_len is not None:
                    ali_prev = ali_prev.masked_fill(pad_mask, 0) / enc_len[..., None]
                else:
                    ali_prev = ali_prev / T
            att = F.conv1d(ali_prev[:, None], sd[p + "F.weight"], sd[p + "F.bias"],
                           padding=loc_context)
            att = F.conv1d(att, sd[p + "att.weight"])
            s = s + att.transpose(1, 2)
        score = F.linear(torch.tanh(s), sd[p + "w.weight"]).squeeze(-1)
    if enc_len is not None:
        score = score.masked_fill(pad_mask, float("-inf"))
    ali = torch.softmax(score, -1)
    return ali, torch.sum(ali[..., None] * enc_pad, 1)


def multi_head_step(sd, p, kind, enc_pad, cache, pad_mask, enc_len, dec_prev, ali_prev, heads,
                    scaled=True, loc_context=0):
    """MHCtx / MHDot / MHLocAttention.forward (attention.py:266-531) restated head by head: head h
    is a single-head attention on columns h A .. (h + 1) A of key_proj / enc_proj / dec_proj, on
    its own slices of the grouped convolutions `w`, `F`, `att`; the concatenated head contexts go
    through ctx_proj.  -> (ali N x H x T, ctx N x D_enc)"""
    N, T, _ = enc_pad.shape
    if "key" not in cache:
        cache["key"] = F.linear(enc_pad, sd[p + "key_proj.weight"])
        cache["val"] = F.linear(enc_pad, sd[p + "enc_proj.weight"], sd.get(p + "enc_proj.bias"))
    key, val = cache["key"], cache["val"]
    A = key.shape[-1] // heads
    query = F.linear(dec_prev, sd[p + "dec_proj.weight"], sd.get(p + "dec_proj.bias"))
    alis, ctxs = [], []
    for h in range(heads):
        k, v, q = (x[..., h * A:(h + 1) * A] for x in (key, val, query))
        if kind == "mhdot":
            score = torch.einsum("nta,na->nt", k, q)
            if scaled:
                score = score / (A**0.5)
        else:
            s = k + q[:, None]
            if kind == "mhloc":
                C = sd[p + "F.weight"].shape[0] // heads
                if ali_prev is None:
                    prev = torch.ones(N, T)
                    prev = prev / T if enc_len is None else \
                        prev.masked_fill(pad_mask, 0) / enc_len[:, None]
                else:
                    prev = ali_prev[:, h]
                loc = F.conv1d(prev[:, None], sd[p + "F.weight"][h * C:(h + 1) * C],
                               sd[p + "F.bias"][h * C:(h + 1) * C], padding=loc_context)
                s = s + F.conv1d(loc, sd[p + "att.weight"][h * A:(h + 1) * A]).transpose(1, 2)
            score = torch.tanh(s) @ sd[p + "w.weight"][h, :, 0]
        if enc_len is not None:
            score = score.masked_fill(pad_mask, float("-inf"))
        ali = torch.softmax(score, -1)
        alis.append(ali)
        ctxs.append(torch.einsum("nt,nta->na", ali, v))
    ctx = F.linear(torch.cat(ctxs, -1), sd[p + "ctx_proj.weight"], sd[p + "ctx_proj.bias"])
    return torch.stack(alis, 1), ctx


def rnn_cell(sd, p, rnn, x, h, c):
    """one time step of a single nn.LSTM / nn.GRU / nn.RNN(tanh) layer whose parameters are
    sd[p + "weight_ih"] ... (torch's gate order: LSTM i | f | g | o, GRU r | z | n; a projected LSTM
    emits and feeds back h W_hr^T) -> (h, c)"""
    gx = F.linear(x, sd[p + "weight_ih"], sd.get(p + "bias_ih"))
    gh = F.linear(h, sd[p + "weight_hh"], sd.get(p + "bias_hh"))
    if rnn == "lstm":
        gi, gf, gg, go = (gx + gh).chunk(4, -1)
        c = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
        h = torch.sigmoid(go) * torch.tanh(c)
        if p + "weight_hr" in sd:
            h = F.linear(h, sd[p + "weight_hr"])
        return h, c
    if rnn == "gru":
        xr, xz, xn = gx.chunk(3, -1)
        hr, hz, hn = gh.chunk(3, -1)
        r, z = torch.sigmoid(xr + hr), torch.sigmoid(xz + hz)
        n = torch.tanh(xn + r * hn)
        return (1 - z) * n + z * h, c
    if rnn == "rnn_tanh":
        return torch.tanh(gx + gh), c
    raise ValueError(rnn)


def rnn_att_decoder(sd, enc_pad, enc_len, tgt_pad, kind, num_layers, input_feeding=False,
                    scaled=True, loc_context=0, att_prefix="att_net.", dec_prefix="decoder.",
                    heads=1, schedule_sampling=0.0, rnn="lstm", add_ln=False, onehot_embed=False):
    """TorchRNNDecoder.forward (decoder.py:69-218), teacher forced or with scheduled sampling (one
    `random.random()` draw per step t > 0, like the reference) -> (outs N x To x V, alis N x To x T).
    rnn: "lstm" (with `weight_hr` in the state dict: projected) | "gru" | "rnn_tanh"; add_ln: the
    LayerNormRNN wrapper (decoder.py:18-66: per-layer single-layer RNNs `decoder.rnns.i`, a LayerNorm
    `decoder.norm.i` behind each); onehot_embed: the previous token enters as its one-hot code"""
    import random
    N, T, D = enc_pad.shape
    a, d = att_prefix, dec_prefix
    enc_part = F.linear(enc_pad, sd[a + "enc_proj.weight"], sd.get(a + "enc_proj.bias"))
    mh_cache = {}
    pad_mask = None if enc_len is None else torch.arange(T)[None, :] >= enc_len[:, None]

    def layer_prefix(l):  # parameters of layer l, up to the "_l{k}" suffix torch appends
        return (d + f"decoder.rnns.{l}.", "_l0") if add_ln else (d + "decoder.", f"_l{l}")

    def params_of(l):
        pre, suf = layer_prefix(l)
        return {k: sd[pre + k + suf] for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh", "weight_hr")
                if pre + k + suf in sd}

    layers = [params_of(l) for l in range(num_layers)]
    H = layers[0]["weight_hh"].shape[1]           # width of the fed-back state (the projection's if any)
    Hc = layers[0]["weight_hh"].shape[0] // {"lstm": 4, "gru": 3, "rnn_tanh": 1}[rnn]
    h = [torch.zeros(N, H) for _ in range(num_layers)]
    c = [torch.zeros(N, Hc) for _ in range(num_layers)]
    V = sd[d + "pred.weight"].shape[0]
    att_ctx, proj, ali = torch.zeros(N, D), torch.zeros(N, D), None
    outs, alis = [], []
    for t in range(tgt_pad.shape[1]):
        tok = tgt_pad[:, t]
        if t and random.random() < schedule_sampling:
            tok = torch.argmax(outs[-1].detach(), dim=1)
        emb = F.one_hot(tok, V).float() if onehot_embed else F.embedding(tok, sd[d + "vocab_embed.weight"])
        x = torch.cat([emb, proj if input_feeding else att_ctx], -1)
        for l in range(num_layers):  # an RNN on a length-1 sequence with carried state
            h[l], c[l] = rnn_cell(layers[l], "", rnn, x, h[l], c[l])
            x = h[l]
            if add_ln:  # (dropout between the layers is the identity in eval mode)
                x = F.layer_norm(x, (x.shape[-1],), sd[d + f"decoder.norm.{l}.weight"],
                                 sd[d + f"decoder.norm.{l}.bias"])
        if kind.startswith("mh"):
            ali, att_ctx = multi_head_step(sd, a, kind, enc_pad, mh_cache, pad_mask, enc_len, x,
                                           ali, heads, scaled, loc_context)
        else:
            ali, att_ctx = attention_step(sd, a, kind, enc_pad, enc_part, pad_mask, enc_len, x,
                                          ali, scaled, loc_context)
        proj = torch.relu(F.linear(torch.cat([x, att_ctx], -1), sd[d + "proj.weight"],
                                   sd[d + "proj.bias"]))
        outs.append(F.linear(proj, sd[d + "pred.weight"], sd[d + "pred.bias"]))
        alis.append(ali)
    return torch.stack(outs, 1), torch.stack(alis, 1)
