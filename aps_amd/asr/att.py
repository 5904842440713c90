"""
Encoder-decoder acoustic models: `asr@att` (attention + RNN decoder) and `asr@xfmr` (Transformer
decoder) of aps/asr/att.py:20-274 (forward path: asr_transform -> encoder -> [CTC branch] ->
teacher-forced decoder).  Beam / greedy search and rescoring are decoding-side control flow
outside the hot path (SURVEY.md 8 "out of scope").
"""
from typing import Dict, Optional

import torch as th
import torch.nn as nn

from aps_amd.asr.base.attention import att_instance
from aps_amd.asr.base.decoder import TorchRNNDecoder
from aps_amd.asr.ctc import AMForwardType, ASREncoderBase, NoneOrTensor
from aps_amd.asr.transformer.decoder import TorchTransformerDecoder
from aps_amd.libs import ApsRegisters


class ASREncoderDecoderBase(ASREncoderBase):
    """encoder + decoder models (att.py:20-46)"""

    def __init__(self, input_size: int, vocab_size: int, sos: int = -1, eos: int = -1,
                 ctc: bool = False, asr_transform: Optional[nn.Module] = None,
                 enc_type: str = "pytorch_rnn", enc_proj: int = -1,
                 enc_kwargs: Optional[Dict] = None) -> None:
        super(ASREncoderDecoderBase, self).__init__(input_size, vocab_size, ctc=ctc, ead=True,
                                                    asr_transform=asr_transform,
                                                    enc_type=enc_type, enc_proj=enc_proj,
                                                    enc_kwargs=enc_kwargs)
        if eos < 0 or sos < 0:
            raise RuntimeError(f"Unsupported SOS/EOS value: {sos}/{eos}")
        self.sos = sos
        self.eos = eos


@ApsRegisters.asr.register("asr@att")
class AttASR(ASREncoderDecoderBase):
    """(Non-)Transformer encoder + attention + RNN decoder (att.py:49-118)"""

    def __init__(self, input_size: int = 80, vocab_size: int = 30, sos: int = -1, eos: int = -1,
                 ctc: bool = False, asr_transform: Optional[nn.Module] = None,
                 att_type: str = "ctx", att_kwargs: Dict = {}, enc_type: str = "common",
                 dec_type: str = "rnn", enc_proj: int = -1, enc_kwargs: Dict = {},
                 dec_dim: int = 512, dec_kwargs: Dict = {}) -> None:
        super(AttASR, self).__init__(input_size, vocab_size, sos=sos, eos=eos, ctc=ctc,
                                     asr_transform=asr_transform, enc_type=enc_type,
                                     enc_proj=enc_proj, enc_kwargs=enc_kwargs)
        if dec_type != "rnn":
            raise ValueError("AttASR: currently decoder must be rnn")
        if self.is_xfmr_encoder:
            enc_proj = enc_kwargs["arch_kwargs"]["att_dim"]
        self.att_net = att_instance(att_type, enc_proj, dec_dim, **att_kwargs)
        self.decoder = TorchRNNDecoder(enc_proj, vocab_size - 1 if ctc else vocab_size, **dec_kwargs)

    def forward(self, x_pad: th.Tensor, x_len: NoneOrTensor, y_pad: th.Tensor,
                y_len: NoneOrTensor, ssr: float = 0) -> AMForwardType:
        """x_pad N x Ti x D | N x S, y_pad N x To (starting with sos) ->
        (dec_out N x To x V, enc_ctc N x T x V | enc_out, enc_len)"""
        self.att_net.clear()
        enc_out, enc_ctc, enc_len = self._training_prep(x_pad, x_len)
        dec_out, _ = self.decoder(self.att_net, enc_out, enc_len, y_pad, schedule_sampling=ssr)
        return dec_out, enc_ctc, enc_len


@ApsRegisters.asr.register("asr@xfmr")
class XfmrASR(ASREncoderDecoderBase):
    """(Non-)Transformer encoder + Transformer decoder (att.py:216-274)"""

    def __init__(self, input_size: int, vocab_size: int, sos: int = -1, eos: int = -1,
                 ctc: bool = False, asr_transform: Optional[nn.Module] = None,
                 enc_type: str = "xfmr", dec_type: str = "xfmr", enc_proj: int = -1,
                 enc_kwargs: Dict = {}, dec_kwargs: Dict = {}) -> None:
        super(XfmrASR, self).__init__(input_size, vocab_size, sos=sos, eos=eos, ctc=ctc,
                                      asr_transform=asr_transform, enc_type=enc_type,
                                      enc_proj=enc_proj, enc_kwargs=enc_kwargs)
        if dec_type != "xfmr":
            raise ValueError("XfmrASR: currently decoder must be xfmr")
        att_dim = dec_kwargs["arch_kwargs"]["att_dim"]
        if not self.is_xfmr_encoder and enc_proj != att_dim:
            raise ValueError(f"enc_proj({enc_proj}) should be equal to att_dim({att_dim})")
        self.decoder = TorchTransformerDecoder(vocab_size - 1 if ctc else vocab_size, **dec_kwargs)

    def forward(self, x_pad: th.Tensor, x_len: NoneOrTensor, y_pad: th.Tensor,
                y_len: NoneOrTensor, ssr: float = 0) -> AMForwardType:
        """x_pad N x Ti x D | N x S, y_pad N x To (starting with sos) ->
        (dec_out N x To x V, enc_ctc N x T x V | enc_out, enc_len)"""
        enc_out, enc_ctc, enc_len = self._training_prep(x_pad, x_len)
        dec_out = self.decoder(enc_out, enc_len, y_pad, y_len)
        return dec_out, enc_ctc, enc_len
