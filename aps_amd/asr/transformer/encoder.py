"""
TransformerEncoder (aps/asr/transformer/encoder.py:18-106): projection -> positional encoding ->
encoder layers -> optional output projection, same constructor and parameter names.
Built: arch "xfmr" | "cfmr", pose "abs" | "rel" | "xl", proj "conv2d" | "conv1d" | "linear" |
"none", context windows (lctx / rctx / chunk_size).
"""
from typing import Dict, Optional

import torch as th
import torch.nn as nn

from aps_amd.asr.base.encoder import EncRetType
from aps_amd.asr.transformer.impl import get_xfmr_encoder
from aps_amd.asr.transformer.pose import get_xfmr_pose
from aps_amd.asr.transformer.proj import get_xfmr_proj
from aps_amd.nn_ops import linear


class TransformerEncoder(nn.Module):

    def __init__(self,
                 arch: str,
                 input_size: int,
                 output_proj: int = -1,
                 num_layers: int = 6,
                 lctx: int = -1,
                 rctx: int = -1,
                 chunk_size: int = 1,
                 proj: str = "conv2d",
                 proj_kwargs: Dict = {},
                 pose: str = "abs",
                 pose_kwargs: Dict = {},
                 arch_kwargs: Dict = {}):
        super(TransformerEncoder, self).__init__()
        if pose not in ("abs", "rel", "xl"):
            raise NotImplementedError(f"aps_amd encoder: pose '{pose}' is not built (abs|rel|xl)")
        att_dim = arch_kwargs["att_dim"]
        self.proj = None if proj == "none" else get_xfmr_proj(proj, input_size, att_dim,
                                                              **proj_kwargs)
        self.pose = get_xfmr_pose(pose, att_dim // arch_kwargs["nhead"] if pose == "rel" else att_dim,
                                  **pose_kwargs)
        self.pose_type = pose
        self.encoder = get_xfmr_encoder(arch, self.pose_type, num_layers, dict(arch_kwargs))
        self.lctx, self.rctx = lctx, rctx
        self.chunk_size = chunk_size
        self.outp = nn.Linear(att_dim, output_proj) if output_proj > 0 else None

    def forward(self, inp_pad: th.Tensor, inp_len: Optional[th.Tensor]) -> EncRetType:
        """N x Ti x F, N | None -> (N x To x D, N | None)"""
        if self.proj is None:
            enc_inp = inp_pad
        else:
            enc_inp, inp_len = self.proj(inp_pad, inp_len)
        rel = None
        if self.pose_type == "abs":
            enc_inp = self.pose.add(enc_inp)
        else:
            rel = self.pose.table(enc_inp.shape[1])
        window = None
        if self.lctx != -1 or self.rctx != -1:  # prep_context_mask(nframes, chunk, lctx, rctx)
            window = (self.chunk_size, self.lctx, self.rctx)
        enc_out = self.encoder.run(enc_inp, inp_len, rel=rel, window=window)
        if self.outp is not None:
            enc_out = linear(enc_out, self.outp.weight, self.outp.bias)
        return enc_out, inp_len
