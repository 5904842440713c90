"""
RNN attention decoder (aps/asr/base/decoder.py:18-218): embedding (or one-hot code) of the previous
token -> RNN stack step (carried state) -> attention over the encoder output -> projection ->
prediction, one target position after the other.  Every projection is an `aps_linear` launch (bias /
ReLU / residual fused), the cell update `aps_lstm_cell` (plain LSTM) or `aps_rnn_step` (GRU, tanh RNN,
projected LSTM), the per-layer LayerNorm of `LayerNormRNN` `aps_layernorm`, the attention
`aps_att_step`; parameters keep the reference's names (`vocab_embed`, `decoder.weight_ih_l0` ... or
`decoder.rnns.0.weight_ih_l0` / `decoder.norm.0.weight` with add_ln, `proj`, `pred`), the carried state
keeps the reference's structure (nn.LSTM: (h, c); nn.GRU / nn.RNN: h; LayerNormRNN: a list per layer).

Built: rnn = "lstm" | "gru" | "rnn_tanh" ("rnn_relu" cannot be constructed in the reference either:
component.py:158 maps it to nn.ReLU), add_ln, proj_size (LSTM), onehot_embed, teacher forcing and
scheduled sampling, `input_feeding` on / off; train() mode applies the dropouts (between the layers
and on the projection).  Under autograd (round 5) the cell steps run on grad_ops.RnnCellFn (aps_rnn_step /
aps_rnn_step_backward), every projection and LayerNorm on their HIP adjoints, and the attention step's scores /
softmax / context on torch's own differentiable ops (asr/base/attention.py: a documented torch fall-through of
the TRAINING path; inference stays on aps_att_step).
"""
import random
from typing import List, Optional, Tuple, Union

import torch as th
import torch.nn as nn

from aps_amd import _native as nat
from aps_amd.grad_ops import DropoutFn, draw_seed, dropout
from aps_amd.nn_ops import RNN_STEP_MODES, layernorm, linear

HiddenType = Union[th.Tensor, Tuple[th.Tensor, th.Tensor], list]


def lstm_cell(pre: th.Tensor, c_prev: Optional[th.Tensor]) -> Tuple[th.Tensor, th.Tensor]:
    """gate pre-activations N x 4H (+ previous cell N x H) -> (h, c)"""
    nat.require_device(pre, c_prev)
    N, H4 = pre.shape
    H = H4 // 4
    h = th.empty(N, H, device=pre.device, dtype=th.float32)
    c = th.empty(N, H, device=pre.device, dtype=th.float32)
    rc = nat.load().aps_lstm_cell(nat.ptr(nat.f32c(pre)),
                                  nat.ptr(None if c_prev is None else nat.f32c(c_prev)), nat.ptr(h),
                                  nat.ptr(c), N, H, nat.stream_of(pre))
    nat.check(rc, "aps_lstm_cell")
    return h, c


def torch_rnn(mode: str, input_size: int, hidden_size: int, num_layers: int = 1, bias: bool = True,
              dropout: float = 0., proj_size: int = -1, bidirectional: bool = False) -> nn.Module:
    """PyTorchRNN (component.py:145-189): the nn.LSTM / nn.GRU / nn.RNN that holds the parameters"""
    mode = mode.upper()
    kwargs = dict(bias=bias, dropout=dropout, batch_first=True, bidirectional=bidirectional)
    if mode == "LSTM":
        if proj_size > 0:
            kwargs["proj_size"] = proj_size
        return nn.LSTM(input_size, hidden_size, num_layers, **kwargs)
    if mode == "GRU":
        return nn.GRU(input_size, hidden_size, num_layers, **kwargs)
    if mode == "RNN_TANH":
        return nn.RNN(input_size, hidden_size, num_layers, nonlinearity="tanh", **kwargs)
    if mode == "RNN_RELU":
        raise ValueError("RNN_RELU: the reference maps it to nn.ReLU (component.py:158) and cannot build it")
    raise ValueError(f"Unsupported RNNs: {mode}")


def rnn_cell_step(rnn: nn.RNNBase, layer: int, x: th.Tensor, h_prev: Optional[th.Tensor],
                  c_prev: Optional[th.Tensor]):
    """one time step of layer `layer` of an nn.LSTM / nn.GRU / nn.RNN with carried state:
    x N x D_in, h_prev N x H_out | None, c_prev N x H | None -> (h N x H_out, c N x H | None)"""
    mode = RNN_STEP_MODES[rnn.mode]
    G = {0: 3, 1: 1, 2: 1, 3: 4}[mode]
    H = rnn.hidden_size
    P = rnn.proj_size if getattr(rnn, "proj_size", 0) > 0 else 0
    sfx = f"_l{layer}"
    w_ih, w_hh = getattr(rnn, "weight_ih" + sfx), getattr(rnn, "weight_hh" + sfx)
    b_ih = getattr(rnn, "bias_ih" + sfx) if rnn.bias else None
    b_hh = getattr(rnn, "bias_hh" + sfx) if rnn.bias else None
    N = x.shape[0]
    if nat.needs_grad(x, h_prev, c_prev, w_ih, w_hh, b_ih, b_hh):
        # under autograd (round 5: the RNN attention decoder trains): both products on `linear` (its adjoint
        # carries the gradients to x, h_prev and the weights), the cell on RnnCellFn (aps_rnn_step /
        # aps_rnn_step_backward), the projection of a projected LSTM on `linear` again
        from aps_amd.grad_ops import RnnCellFn
        if h_prev is None:
            h_prev = th.zeros(N, P if P else H, device=x.device, dtype=th.float32)
        if mode == 3 and c_prev is None:
            c_prev = th.zeros(N, H, device=x.device, dtype=th.float32)
        gx = linear(x, w_ih, b_ih)
        gh = linear(h_prev, w_hh, b_hh)
        h_new, c_new = RnnCellFn.apply(gx, gh, None if P else h_prev, c_prev, mode, H)
        if P:
            h_new = linear(h_new, getattr(rnn, "weight_hr" + sfx))
        return h_new, c_new
    gx = linear(x, w_ih, b_ih)
    nat.require_device(x, h_prev, c_prev)
    if mode == 3 and not P:  # the plain LSTM cell: the recurrent product rides in as the residual
        if h_prev is None:
            pre = gx if b_hh is None else gx + b_hh
        else:
            pre = linear(h_prev, w_hh, b_hh, residual=gx)
        return lstm_cell(pre, c_prev)
    if h_prev is None:
        h_prev = th.zeros(N, P if P else H, device=x.device, dtype=th.float32)
    if mode == 3 and c_prev is None:
        c_prev = th.zeros(N, H, device=x.device, dtype=th.float32)
    gh = linear(h_prev, w_hh, b_hh)  # N x G H (the GRU keeps r * (W_hn h + b_hn) apart: not summed here)
    h_new = th.empty(N, H, device=x.device, dtype=th.float32)
    c_new = th.empty(N, H, device=x.device, dtype=th.float32) if mode == 3 else None
    # (a projected LSTM's cell sees no h_prev of its own width: the kernel only needs it to freeze rows
    # past a length, and a decoder step has none)
    rc = nat.load().aps_rnn_step(nat.ptr(nat.f32c(gx)), G * H, nat.ptr(gh),
                                 nat.ptr(None if P else nat.f32c(h_prev)),
                                 nat.ptr(None if c_prev is None else nat.f32c(c_prev)), nat.ptr(None), 0,
                                 nat.ptr(h_new), nat.ptr(c_new), nat.ptr(None), 0, N, H, mode,
                                 nat.stream_of(x))
    nat.check(rc, "aps_rnn_step")
    if P:
        h_new = linear(h_new, getattr(rnn, "weight_hr" + sfx))
    return h_new, c_new


class OneHotEmbedding(nn.Module):
    """Onehot embedding layer (component.py:58-82)"""

    def __init__(self, vocab_size: int):
        super(OneHotEmbedding, self).__init__()
        self.vocab_size = vocab_size

    def extra_repr(self) -> str:
        return f"vocab_size={self.vocab_size}"

    def forward(self, x: th.Tensor) -> th.Tensor:
        """... -> ... x V"""
        H = th.zeros(list(x.shape) + [self.vocab_size], dtype=th.float32, device=x.device)
        return H.scatter(-1, x[..., None], 1)


class LayerNormRNN(nn.Module):
    """RNNs with layer normalization (decoder.py:18-66): one single-layer RNN per layer, dropout between
    the layers, a LayerNorm behind every layer; the carried state is the list of the layers' states"""

    def __init__(self, mode: str, input_size: int, hidden_size: int, proj_size: int = -1,
                 num_layers: int = 1, bias: bool = True, dropout: float = 0.,
                 bidirectional: bool = False) -> None:
        super(LayerNormRNN, self).__init__()
        if bidirectional:
            raise NotImplementedError("aps_amd LayerNormRNN: unidirectional (its only use is the decoder)")
        inner_size = proj_size if proj_size > 0 else hidden_size
        self.rnns = nn.ModuleList([
            torch_rnn(mode, inner_size if i else input_size, hidden_size, num_layers=1,
                      proj_size=proj_size, bias=bias, dropout=0, bidirectional=False)
            for i in range(num_layers)
        ])
        self.dropout = nn.ModuleList(nn.Dropout(p=dropout) for _ in range(num_layers - 1))
        self.norm = nn.ModuleList(nn.LayerNorm(inner_size) for _ in range(num_layers))

    def step(self, x: th.Tensor, hx: Optional[list]):
        """x N x D -> (out N x D_dec, [state of every layer, shaped like the single-layer RNN's])"""
        ret = []
        for i, rnn in enumerate(self.rnns):
            hid = None if hx is None else hx[i]
            if rnn.mode == "LSTM":
                h_prev, c_prev = (None, None) if hid is None else (hid[0][0], hid[1][0])
            else:
                h_prev, c_prev = (None if hid is None else hid[0]), None
            x, c = rnn_cell_step(rnn, 0, x, h_prev, c_prev)
            ret.append((x[None], c[None]) if rnn.mode == "LSTM" else x[None])
            if i != len(self.rnns) - 1:
                x = dropout(x, self.dropout[i])
            n = self.norm[i]
            x = layernorm(x, n.weight, n.bias, n.eps)
        return x, ret


class TorchRNNDecoder(nn.Module):
    """PyTorch's RNN decoder (decoder.py:69-218)"""

    def __init__(self, enc_proj: int, vocab_size: int, rnn: str = "lstm", add_ln: bool = False,
                 num_layers: int = 3, proj_size: int = -1, hidden: int = 512, dropout: float = 0.0,
                 input_feeding: bool = False, onehot_embed: bool = False) -> None:
        super(TorchRNNDecoder, self).__init__()
        if not onehot_embed:
            self.vocab_embed = nn.Embedding(vocab_size, hidden)
            input_size = enc_proj + hidden
        else:
            self.vocab_embed = OneHotEmbedding(vocab_size)
            input_size = enc_proj + vocab_size
        if add_ln:
            self.decoder = LayerNormRNN(rnn, input_size, hidden, proj_size=proj_size,
                                        num_layers=num_layers, dropout=dropout, bidirectional=False)
        else:
            self.decoder = torch_rnn(rnn, input_size, hidden, proj_size=proj_size,
                                     num_layers=num_layers, dropout=dropout, bidirectional=False)
        self.proj = nn.Linear((proj_size if proj_size > 0 else hidden) + enc_proj, enc_proj)
        self.drop = nn.Dropout(p=dropout)
        self.pred = nn.Linear(enc_proj, vocab_size)
        self.input_feeding = input_feeding
        self.vocab_size = vocab_size

    def step_decoder(self, emb_pre: th.Tensor, att_ctx: th.Tensor,
                     dec_hid: Optional[HiddenType] = None) -> Tuple[th.Tensor, HiddenType]:
        """emb_pre N x D_emb, att_ctx N x D_enc -> (dec_out N x D_dec, state: (h L x N x H_out,
        c L x N x H) of an nn.LSTM, h L x N x H of an nn.GRU / nn.RNN, a list per layer with add_ln)"""
        x = th.cat([emb_pre, att_ctx], dim=-1)
        if isinstance(self.decoder, LayerNormRNN):
            return self.decoder.step(x, dec_hid)
        rnn = self.decoder
        lstm = rnn.mode == "LSTM"
        hs, cs = [], []
        for layer in range(rnn.num_layers):
            if dec_hid is None:
                h_prev = c_prev = None
            elif lstm:
                h_prev, c_prev = dec_hid[0][layer], dec_hid[1][layer]
            else:
                h_prev, c_prev = dec_hid[layer], None
            x, c = rnn_cell_step(rnn, layer, x, h_prev, c_prev)
            hs.append(x)
            cs.append(c)
            if rnn.training and rnn.dropout > 0 and layer + 1 < rnn.num_layers:
                x = DropoutFn.apply(x, rnn.dropout, draw_seed())  # nn.LSTM / nn.GRU: between the layers
        return x, ((th.stack(hs), th.stack(cs)) if lstm else th.stack(hs))

    def step(self, att_net: nn.Module, out_pre: th.Tensor, enc_out: th.Tensor, att_ctx: th.Tensor,
             dec_hid: Optional[HiddenType] = None, att_ali: Optional[th.Tensor] = None,
             proj: Optional[th.Tensor] = None, enc_len: Optional[th.Tensor] = None):
        """one prediction step (decoder.py:137-165) -> (pred, att_ctx, dec_hid, att_ali, proj)"""
        emb_pre = self.vocab_embed(out_pre)  # row gather / one-hot code
        dec_out, dec_hid = self.step_decoder(emb_pre, proj if self.input_feeding else att_ctx,
                                             dec_hid=dec_hid)
        att_ali, att_ctx = att_net(enc_out, enc_len, dec_out, att_ali)
        proj = dropout(linear(th.cat([dec_out, att_ctx], dim=-1), self.proj.weight, self.proj.bias,
                              act="relu"), self.drop)
        pred = linear(proj, self.pred.weight, self.pred.bias)
        return pred, att_ctx, dec_hid, att_ali, proj

    def forward(self, att_net: nn.Module, enc_pad: th.Tensor, enc_len: Optional[th.Tensor],
                tgt_pad: th.Tensor, schedule_sampling: float = 0) -> Tuple[th.Tensor, th.Tensor]:
        """enc_pad N x Ti x D_enc, tgt_pad N x To -> (outs N x To x V, alis N x To x Ti)"""
        N, _, D_enc = enc_pad.shape
        outs: List[th.Tensor] = []
        alis: List[th.Tensor] = []
        att_ali, dec_hid = None, None
        att_ctx = th.zeros([N, D_enc], device=enc_pad.device)
        proj = th.zeros([N, D_enc], device=enc_pad.device)
        for t in range(tgt_pad.shape[-1]):
            # scheduled sampling (decoder.py:196-200): with probability `schedule_sampling` the
            # previous PREDICTION is fed back instead of the ground truth; the draw is Python's
            # `random`, consumed exactly like the reference does (one draw per step t > 0)
            if t and random.random() < schedule_sampling:
                tok_pre = th.argmax(outs[-1].detach(), dim=1)
            else:
                tok_pre = tgt_pad[:, t]
            pred, att_ctx, dec_hid, att_ali, proj = self.step(att_net, tok_pre, enc_pad,
                                                              att_ctx, dec_hid=dec_hid,
                                                              att_ali=att_ali, enc_len=enc_len,
                                                              proj=proj)
            outs.append(pred)
            alis.append(att_ali)
        return th.stack(outs, dim=1), th.stack(alis, dim=1)
