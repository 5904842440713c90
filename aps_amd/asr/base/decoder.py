"""
RNN attention decoder (aps/asr/base/decoder.py:69-218): embedding of the previous token ->
LSTM stack step (carried state) -> attention over the encoder output -> projection -> prediction,
one target position after the other.  Every projection is an `aps_linear` launch (bias / ReLU /
residual fused), the cell update `aps_lstm_cell`, the attention `aps_att_step`; parameters keep
the reference's names (`vocab_embed`, `decoder.weight_ih_l0` ..., `proj`, `pred`).

Built: rnn = "lstm" (no projection, no layer norm), teacher forcing (schedule_sampling = 0),
`input_feeding` on / off; train() mode applies the dropouts (between the LSTM layers and on the
projection) but the cell / attention steps have no backward kernels.
"""
import random
from typing import List, Optional, Tuple

import torch as th
import torch.nn as nn

from aps_amd import _native as nat
from aps_amd.grad_ops import DropoutFn, draw_seed, dropout
from aps_amd.nn_ops import linear

HiddenType = Tuple[th.Tensor, th.Tensor]


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


class TorchRNNDecoder(nn.Module):
    """PyTorch's RNN decoder (decoder.py:69-218)"""

    def __init__(self, enc_proj: int, vocab_size: int, rnn: str = "lstm", add_ln: bool = False,
                 num_layers: int = 3, proj_size: int = -1, hidden: int = 512, dropout: float = 0.0,
                 input_feeding: bool = False, onehot_embed: bool = False) -> None:
        super(TorchRNNDecoder, self).__init__()
        if rnn.lower() != "lstm" or add_ln or proj_size > 0 or onehot_embed:
            raise NotImplementedError("aps_amd RNN decoder: lstm cells without layer norm / "
                                      "projection / one-hot embedding only")
        self.vocab_embed = nn.Embedding(vocab_size, hidden)
        self.decoder = nn.LSTM(enc_proj + hidden, hidden, num_layers=num_layers, batch_first=True,
                               dropout=dropout, bidirectional=False)
        self.proj = nn.Linear(hidden + enc_proj, enc_proj)
        self.drop = nn.Dropout(p=dropout)
        self.pred = nn.Linear(enc_proj, vocab_size)
        self.input_feeding = input_feeding
        self.vocab_size = vocab_size

    def step_decoder(self, emb_pre: th.Tensor, att_ctx: th.Tensor,
                     dec_hid: Optional[HiddenType] = None) -> Tuple[th.Tensor, HiddenType]:
        """emb_pre N x D_emb, att_ctx N x D_enc -> (dec_out N x H, (h L x N x H, c L x N x H))"""
        rnn = self.decoder
        x = th.cat([emb_pre, att_ctx], dim=-1)
        hs, cs = [], []
        for layer in range(rnn.num_layers):
            w_ih, w_hh = getattr(rnn, f"weight_ih_l{layer}"), getattr(rnn, f"weight_hh_l{layer}")
            b_ih = getattr(rnn, f"bias_ih_l{layer}") if rnn.bias else None
            b_hh = getattr(rnn, f"bias_hh_l{layer}") if rnn.bias else None
            pre = linear(x, w_ih, b_ih)
            if dec_hid is None:  # zero state: only the recurrent bias contributes
                if b_hh is not None:
                    pre = pre + b_hh
                c_prev = None
            else:
                pre = linear(dec_hid[0][layer], w_hh, b_hh, residual=pre)
                c_prev = dec_hid[1][layer]
            x, c = lstm_cell(pre, c_prev)
            hs.append(x)
            if rnn.training and rnn.dropout > 0 and layer + 1 < rnn.num_layers:
                x = DropoutFn.apply(x, rnn.dropout, draw_seed())  # nn.LSTM: between the layers
            cs.append(c)
        return x, (th.stack(hs), th.stack(cs))

    def step(self, att_net: nn.Module, out_pre: th.Tensor, enc_out: th.Tensor, att_ctx: th.Tensor,
             dec_hid: Optional[HiddenType] = None, att_ali: Optional[th.Tensor] = None,
             proj: Optional[th.Tensor] = None, enc_len: Optional[th.Tensor] = None):
        """one prediction step (decoder.py:137-165) -> (pred, att_ctx, dec_hid, att_ali, proj)"""
        emb_pre = th.nn.functional.embedding(out_pre, self.vocab_embed.weight)  # row gather
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
