"""
Encoder-side acoustic model: asr_transform -> encoder -> optional CTC projection, the
`ASREncoderBase` / `CtcASR` surface of aps/asr/ctc.py:23-203 (forward path; CTC beam search and
alignment are decoding-side code outside the hot path, SURVEY.md 8 "out of scope").
"""
from typing import Dict, Optional, Tuple

import torch as th
import torch.nn as nn

from aps_amd.asr.base.encoder import BaseEncoder
from aps_amd.asr.transformer.encoder import TransformerEncoder
from aps_amd.libs import ApsRegisters
from aps_amd.nn_ops import linear

NoneOrTensor = Optional[th.Tensor]
AMForwardType = Tuple[th.Tensor, NoneOrTensor, NoneOrTensor]


def encoder_instance(enc_type: str, inp_features: int, out_features: int, enc_kwargs: Dict,
                     encoders=BaseEncoder) -> nn.Module:
    """aps/asr/base/encoder.py:21-51 for the encoder classes built here; "concat" chains several of
    them (enc_kwargs: an ordered {type: kwargs} mapping), every stage but the last keeps its own
    width, the last one projects to out_features"""

    def one(kind, inp, out, **kwargs):
        if kind not in encoders:
            raise RuntimeError(f"Unknown encoder type: {kind}")
        return encoders[kind](inp, out, **kwargs)

    if enc_type != "concat":
        return one(enc_type, inp_features, out_features, **enc_kwargs)
    if len(enc_kwargs) <= 1:
        raise ValueError("Please use >=2 encoders for 'concat' type encoder")
    stages = []
    for i, (kind, kwargs) in enumerate(enc_kwargs.items()):
        last = i == len(enc_kwargs) - 1
        stages.append(one(kind, inp_features if i == 0 else stages[-1].out_features,
                          out_features if last else -1, **kwargs))
    return ConcatEncoder(stages)


class ConcatEncoder(nn.ModuleList):
    """encoders run one after the other, frame lengths handed along (encoder.py:54-72)"""

    def forward(self, inp: th.Tensor, inp_len: Optional[th.Tensor]):
        for encoder in self:
            inp, inp_len = encoder(inp, inp_len)
        return inp, inp_len


class ASREncoderBase(nn.Module):
    """ASR encoder (+ CTC branch) (ctc.py:23-134)"""

    def __init__(self, input_size: int, vocab_size: int, ctc: bool = True, ead: bool = False,
                 asr_transform: Optional[nn.Module] = None, enc_type: str = "pytorch_rnn",
                 enc_proj: int = -1, enc_kwargs: Optional[Dict] = None) -> None:
        super(ASREncoderBase, self).__init__()
        assert ctc or ead
        ctc_only = ctc and not ead
        self.vocab_size = vocab_size
        self.asr_transform = asr_transform
        enc_kwargs = dict(enc_kwargs or {})
        if enc_type in ["xfmr", "cfmr"]:
            self.is_xfmr_encoder = True
            enc_proj = enc_kwargs["arch_kwargs"]["att_dim"]
            enc_kwargs["output_proj"] = vocab_size if ctc_only else -1
            self.encoder = TransformerEncoder(enc_type, input_size, **enc_kwargs)
        else:
            self.is_xfmr_encoder = False
            self.encoder = encoder_instance(enc_type, input_size,
                                            vocab_size if ctc_only else enc_proj, enc_kwargs)
        self.ctc = nn.Linear(enc_proj, vocab_size) if ead and ctc else None

    def _training_prep(self, x_pad: th.Tensor, x_len: NoneOrTensor) -> AMForwardType:
        """N x Ti x D | N x S -> (enc_out N x T x D, enc_ctc N x T x V | enc_out, enc_len)
        (ctc.py:113-134)"""
        if self.asr_transform:
            x_pad, x_len = self.asr_transform(x_pad, x_len)
        enc_out, enc_len = self.encoder(x_pad, x_len)
        enc_ctc = enc_out
        if self.ctc:
            enc_ctc = linear(enc_out, self.ctc.weight, self.ctc.bias)
        return enc_out, enc_ctc, enc_len

    def _decoding_prep(self, x: th.Tensor, batch_first: bool = True) -> th.Tensor:
        """one utterance (S | C x S | T x F) -> encoder output (ctc.py:86-111)"""
        x_dim = x.dim()
        if self.asr_transform:
            if x_dim not in [1, 2]:
                raise RuntimeError("Expect 1/2D (single/multi-channel waveform or single " +
                                   f"channel feature) tensor, but get {x_dim}")
            x, _ = self.asr_transform(x[None, ...], None)
        else:
            if x_dim not in [2, 3]:
                raise RuntimeError("Expect 2/3D (single or multi-channel waveform) " +
                                   f"tensor, but got {x_dim}")
            x = x[None, ...]
        enc_out, _ = self.encoder(x, None)
        return enc_out if batch_first else enc_out.transpose(0, 1)


@ApsRegisters.asr.register("asr@ctc")
class CtcASR(ASREncoderBase):
    """ASR encoder trained with CTC (ctc.py:137-169): forward = _training_prep"""

    def __init__(self, input_size: int = 80, vocab_size: int = 30, ctc: bool = True,
                 ead: bool = False, asr_transform: Optional[nn.Module] = None,
                 enc_type: str = "pytorch_rnn", enc_proj: int = -1,
                 enc_kwargs: Optional[Dict] = None) -> None:
        super(CtcASR, self).__init__(input_size, vocab_size, ctc=ctc, ead=ead,
                                     asr_transform=asr_transform, enc_type=enc_type,
                                     enc_proj=enc_proj, enc_kwargs=enc_kwargs)

    def forward(self, x_pad: th.Tensor, x_len: NoneOrTensor) -> AMForwardType:
        return self._training_prep(x_pad, x_len)
