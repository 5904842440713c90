"""
DCCRN: deep complex convolutional recurrent network (aps/sse/bss/dccrn.py:16-349), forward /
inference path on the MI355X kernels:

  STFT (bin-fastest store N x T x F x 2 == channels-last complex input)  ->  encoder: 7 x one
  launch of the channels-last conv kernel  ->  complex LSTM (4 LSTM passes = 2 batched runs of the
  persistent recurrence kernel, complex combination in the projection GEMMs' alpha / residual)
  ->  decoder: 7 x one launch (skip additions fused)  ->  complex ratio masks + masking in one
  kernel  ->  iSTFT.

State-dict keys are the reference's.

Training (`train()`, cmd/train_ss.py): the same layout with every stage differentiable -- STFT /
iSTFT adjoints (transform/utils.py), the UNet blocks' training form (dcunet.py: convolution and
transposed convolution adjoints, batch-statistics BatchNorm, LeakyReLU), the LSTM stacks' BPTT
(grad_ops.LstmFn; the real / imaginary LSTMs run one after the other there), the projections'
GEMM adjoints and the mask kernel's adjoint (aps_dccrn_mask_backward).
"""
from typing import List, Optional, Tuple, Union

import torch as th
import torch.nn as nn

from aps_amd import _native as nat
from aps_amd.const import EPSILON
from aps_amd.libs import ApsRegisters
from aps_amd.nn_ops import (linear, lstm_forward, lstm_pair_forward, lstm_supported,
                            rnn_step_forward, rnn_step_supported)
from aps_amd.spectrogram import packed_view
from aps_amd.sse.base import MaskNonLinear, SSEBase
from aps_amd.sse.enh.dcunet import Decoder, Encoder, parse_1dstr, parse_2dstr


class LSTMP(nn.Module):
    """LSTM + bias-free projection back to the input width (dccrn.py:16-51)"""

    def __init__(self, in_features: int, hidden_size: int, num_layers: int = 2, dropout: float = 0,
                 bidirectional: bool = False, batch_first: bool = True) -> None:
        super(LSTMP, self).__init__()
        self.lstm = nn.LSTM(in_features, hidden_size, dropout=dropout, num_layers=num_layers,
                            bidirectional=bidirectional, batch_first=batch_first)
        self.proj = nn.Linear(hidden_size * 2 if bidirectional else hidden_size, in_features,
                              bias=False)

    def recur(self, inp: th.Tensor) -> th.Tensor:
        """N x T x D -> N x T x H (the recurrence only)"""
        if lstm_supported(self.lstm, inp):
            return lstm_forward(self.lstm, inp)
        if rnn_step_supported(self.lstm, inp):  # hidden sizes without a persistent kernel
            return rnn_step_forward(self.lstm, inp)
        return self.lstm(inp)[0]  # (autograd / CPU tensors)

    def forward(self, inp: th.Tensor) -> th.Tensor:
        """N x T x C x F -> N x T x C x F"""
        N, T, C, _ = inp.shape
        out = linear(self.recur(inp.reshape(N, T, -1)), self.proj.weight)
        return out.view(N, T, C, -1)


class ComplexLSTMP(nn.Module):
    """(a + bi)(c + di) with LSTMP "multiplications" (dccrn.py:54-94)"""

    def __init__(self, in_features: int, hidden_size: int, num_layers: int = 2, dropout: float = 0,
                 bidirectional: bool = False, batch_first: bool = True) -> None:
        super(ComplexLSTMP, self).__init__()
        kw = dict(num_layers=num_layers, dropout=dropout, bidirectional=bidirectional,
                  batch_first=batch_first)
        self.real = LSTMP(in_features, hidden_size, **kw)
        self.imag = LSTMP(in_features, hidden_size, **kw)

    def run(self, inp_r: th.Tensor, inp_i: th.Tensor) -> Tuple[th.Tensor, th.Tensor]:
        """N x T x D real / imaginary inputs -> N x T x D real / imaginary outputs"""
        N = inp_r.shape[0]
        both = th.cat([inp_r, inp_i], 0)  # each LSTM sees both parts: one batched run per module,
        grad = nat.needs_grad(both, *self.parameters()) or \
            (self.training and self.real.lstm.dropout > 0 and self.real.lstm.num_layers > 1)
        if lstm_supported(self.real.lstm, both) and not self.real.lstm.bidirectional and not grad:
            hr, hi = lstm_pair_forward(self.real.lstm, self.imag.lstm, both)  # both in one launch
        else:
            hr, hi = self.real.recur(both), self.imag.recur(both)
        wr, wi = self.real.proj.weight, self.imag.proj.weight
        # out_r = real(r) - imag(i),  out_i = real(i) + imag(r): the combination is the second
        # projection's alpha / residual epilogue
        out_r = linear(hi[N:], wi, alpha=-1.0, residual=linear(hr[:N], wr))
        out_i = linear(hi[:N], wi, residual=linear(hr[N:], wr))
        return out_r, out_i

    def forward(self, inp: th.Tensor) -> th.Tensor:
        """N x T x C x 2F -> N x T x C x 2F"""
        N, T, C, _ = inp.shape
        inp_r, inp_i = th.chunk(inp, 2, -1)
        out_r, out_i = self.run(inp_r.reshape(N, T, -1), inp_i.reshape(N, T, -1))
        return th.cat([out_r.view(N, T, C, -1), out_i.view(N, T, C, -1)], -1)


class LSTMWrapper(nn.Module):
    """real / complex LSTM over the flattened (channel, frequency) axis (dccrn.py:97-136)"""

    def __init__(self, in_features: int, num_layers: int = 2, dropout: float = 0,
                 hidden_size: int = 512, cplx: bool = True, bidirectional: bool = False) -> None:
        super(LSTMWrapper, self).__init__()
        cls = ComplexLSTMP if cplx else LSTMP
        self.lstm = cls(in_features, hidden_size, dropout=dropout, num_layers=num_layers,
                        bidirectional=bidirectional, batch_first=True)
        self.cplx = cplx

    def run(self, h: th.Tensor) -> th.Tensor:
        """channels-last N x T x F x 2C (real | imag channels; C real channels if not cplx) -> same"""
        N, T, Fd, C2 = h.shape
        if not self.cplx:
            # the reference flattens (channel, frequency) channel-major
            flat = h.permute(0, 1, 3, 2).reshape(N, T, C2 * Fd)
            out = linear(self.lstm.recur(flat), self.lstm.proj.weight)
            return out.view(N, T, C2, Fd).permute(0, 1, 3, 2).contiguous()
        C = C2 // 2
        # the reference flattens (channel, frequency) channel-major
        parts = h.view(N, T, Fd, 2, C).permute(3, 0, 1, 4, 2).reshape(2, N, T, C * Fd)
        out_r, out_i = self.lstm.run(parts[0], parts[1])
        out = th.stack([out_r.view(N, T, C, Fd), out_i.view(N, T, C, Fd)], 3)  # N T C 2 F
        return out.permute(0, 1, 4, 3, 2).reshape(N, T, Fd, C2)

    def forward(self, inp: th.Tensor) -> th.Tensor:
        """N x C x (2)F x T -> N x C x (2)F x T"""
        out = self.lstm(th.einsum("ncft->ntcf", inp))
        return th.einsum("ntcf->ncft", out)


@ApsRegisters.sse.register("sse@dccrn")
class DCCRN(SSEBase):
    """DCCRN (dccrn.py:139-349); K, S, P, O, C as in the reference (frequency, time) order"""

    def __init__(self,
                 cplx: bool = True,
                 K: str = "3,3;3,3;3,3;3,3;3,3;3,3;3,3",
                 S: str = "2,1;2,1;2,1;2,1;2,1;2,1;2,1",
                 P: str = "1,1,1,1,1,1,1",
                 O: str = "0,0,0,0,0,0,0",
                 C: str = "16,32,64,64,128,128,256",
                 num_spks: int = 2,
                 connection: str = "sum",
                 rnn_hidden: int = 512,
                 rnn_layers: int = 2,
                 rnn_resize: int = 1536,
                 rnn_dropout: float = 0,
                 rnn_bidir: bool = False,
                 causal_conv: bool = False,
                 share_decoder: bool = True,
                 enh_transform: Optional[nn.Module] = None,
                 non_linear: str = "tanh",
                 training_mode: str = "time") -> None:
        super(DCCRN, self).__init__(enh_transform, training_mode=training_mode)
        assert enh_transform is not None
        self.cplx = cplx
        self.non_linear = MaskNonLinear(non_linear, enable="all_wo_softmax")
        self.forward_stft = enh_transform.ctx(name="forward_stft")
        self.inverse_stft = enh_transform.ctx(name="inverse_stft")
        K, S = parse_2dstr(K), parse_2dstr(S)
        C, P, O = parse_1dstr(C), parse_1dstr(P), parse_1dstr(O)
        self.encoder = Encoder(cplx, K, S, [1] + C, P, causal=causal_conv)
        if connection == "cat":
            C[-1] *= 2
        heads = [num_spks] if share_decoder else [1] * num_spks
        self.decoder = nn.ModuleList([
            Decoder(cplx, K[::-1], S[::-1], C[::-1] + [n], P[::-1], O[::-1], causal=causal_conv,
                    connection=connection) for n in heads
        ])
        self.rnn = LSTMWrapper(rnn_resize // 2 if cplx else rnn_resize, dropout=rnn_dropout,
                               num_layers=rnn_layers, hidden_size=rnn_hidden,
                               bidirectional=rnn_bidir, cplx=cplx)
        self.num_spks = num_spks
        self.connection = connection
        self.share_decoder = share_decoder

    # ---- kernels' layout ---------------------------------------------------------------------
    def _decode(self, store: th.Tensor, eps: float = EPSILON) -> th.Tensor:
        """STFT store N x T x F x 2 -> decoder output N x T x F x 2S (real | imag mask channels;
        N x T x F x S real masks if not cplx)"""
        if self.cplx:
            inp = store
        else:  # magnitude spectrogram (dccrn.py:259)
            N, T, Fd, _ = store.shape
            inp = th.empty(N, T, Fd, 1, device=store.device, dtype=th.float32)
            rc = nat.load().aps_store_magnitude(nat.ptr(store), nat.ptr(inp), N * T * Fd, float(eps),
                                                nat.stream_of(store))
            nat.check(rc, "aps_store_magnitude")
        enc_h, h = self.encoder.run(inp)
        out_h = self.rnn.run(h)
        if self.connection == "sum":
            h = h + out_h
        elif self.cplx:  # cat([out_h, h]) over complex channels: real parts first
            c = h.shape[-1] // 2
            h = th.cat([out_h[..., :c], h[..., :c], out_h[..., c:], h[..., c:]], -1)
        else:
            h = th.cat([out_h, h], -1)
        skips = enc_h[::-1]
        outs = [dec.run(h, skips) for dec in self.decoder]
        if len(outs) == 1:
            return outs[0]
        if not self.cplx:
            return th.cat(outs, -1)
        # per-speaker decoders: channels (r, i) each -> (r_0 .. r_S-1, i_0 .. i_S-1)
        return th.cat([o[..., :1] for o in outs] + [o[..., 1:] for o in outs], -1)

    def _separate(self, store: th.Tensor, mode: str, eps: float = EPSILON) -> th.Tensor:
        """-> S x N x T x F x 2: the masks (mode "freq") or the masked spectrograms ("time");
        the real-valued network's masks (mode "freq") are S x N x T x F"""
        dec = self._decode(store, eps)
        N, T, Fd, _ = store.shape
        if nat.needs_grad(dec, store):
            from aps_amd.grad_ops import DccrnMaskFn
            return DccrnMaskFn.apply(dec, store if mode == "time" else None, self.num_spks,
                                     self.non_linear.code(), mode == "time", self.cplx, float(eps))
        shape = (self.num_spks, N, T, Fd, 2) if self.cplx or mode == "time" else \
            (self.num_spks, N, T, Fd)
        out = th.empty(*shape, device=store.device, dtype=th.float32)
        rc = nat.load().aps_dccrn_mask(nat.ptr(dec), nat.ptr(store), nat.ptr(out), N * T * Fd,
                                       self.num_spks, self.non_linear.code(),
                                       int(mode == "time"), int(self.cplx), float(eps),
                                       nat.stream_of(store))
        nat.check(rc, "aps_dccrn_mask")
        return out

    def _infer(self, mix: th.Tensor, mode: str = "freq") -> Union[th.Tensor, List[th.Tensor]]:
        nat.require_device(mix)
        store = self.forward_stft.to_store(mix)  # N x T x F x 2
        sep = self._separate(store, mode)
        if mode == "freq" and not self.cplx:
            res = [s.transpose(1, 2) for s in sep]  # N x F x T real masks
        elif mode == "freq":
            res = [packed_view(s) for s in sep]  # N x F x T x 2 like the reference's stack
        else:
            res = [self.inverse_stft(packed_view(s), return_polar=False) for s in sep]
        return res[0] if self.num_spks == 1 else res

    # ---- reference surface --------------------------------------------------------------------
    def infer(self, mix: th.Tensor, mode: str = "time"):
        """S -> S | F x T x 2 per speaker (dccrn.py:296-312)"""
        self.check_args(mix, training=False, valid_dim=[1])
        with th.no_grad():
            sep = self._infer(mix[None, :], mode=mode)
            return sep[0] if self.num_spks == 1 else [s[0] for s in sep]

    def forward(self, s: th.Tensor):
        """N x S -> per speaker N x S (training_mode "time") or N x F x T x 2 ("freq")"""
        self.check_args(s, training=True, valid_dim=[2])
        return self._infer(s, mode=self.training_mode)

    def mask_predict(self, stft: th.Tensor, eps: float = EPSILON) -> th.Tensor:
        """N x T x F x 2 -> (S x) N x T x F x 2 complex masks (dccrn.py:326-349)"""
        nat.require_device(stft)
        masks = self._separate(nat.f32c(stft), "freq", eps)
        return masks[0] if self.num_spks == 1 else masks
