"""Input projection layers before the transformer encoder (aps/asr/transformer/proj.py)."""
from typing import Optional, Tuple

import torch as th
import torch.nn as nn

from aps_amd.asr.base.encoder import Conv2dEncoder
from aps_amd.libs import Register

XfmrProjLayer = Register("xfmr_proj_layer")
ProjOutputType = Tuple[th.Tensor, Optional[th.Tensor]]


def get_xfmr_proj(proj_name: str, in_features: int, att_dim: int, **kwargs) -> nn.Module:
    if proj_name not in XfmrProjLayer:
        raise ValueError(f"Unsupported projection layer: {proj_name}")
    return XfmrProjLayer[proj_name](in_features, att_dim, **kwargs)


@XfmrProjLayer.register("conv2d")
class Conv2dProj(nn.Module):
    """2d-conv subsampling projection (proj.py:105-140); parameters under `conv.`"""

    def __init__(self,
                 input_size: int,
                 embed_dim: int,
                 norm: str = "BN",
                 kernel=3,
                 stride=2,
                 num_layers: int = 2,
                 in_channels: int = 1,
                 conv_channels: int = 256,
                 for_streaming: bool = False) -> None:
        super(Conv2dProj, self).__init__()
        assert num_layers in [2, 3, 4]
        self.conv = Conv2dEncoder(input_size, embed_dim, channel=conv_channels,
                                  in_channels=in_channels, num_layers=num_layers, norm=norm,
                                  kernel=kernel, stride=stride, for_streaming=for_streaming)

    def forward(self, inp: th.Tensor, inp_len: Optional[th.Tensor]) -> ProjOutputType:
        """N x T x F or N x C x T x F -> N x T' x D"""
        return self.conv(inp[:, None] if inp.dim() == 3 else inp, inp_len)
