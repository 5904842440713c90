"""
Spectral-approximation objectives of aps/task/sse.py:207-455 on the HIP front end: the task's STFT
context (`enh_transform.ctx("forward_stft")`, polar output) is the STFT kernel (aps_stft_forward with
polar = 1), the mel projection of MelFreqSaTask the fp32 MFMA GEMM; the element-wise loss arithmetic
and the permutation bookkeeping are torch ops on the task side (aps_amd/task/objf.py).  The network
under the task may be any module with an `enh_transform` attribute that returns masks N x F x T.
"""
from typing import Dict, Optional

import torch as th
import torch.nn as nn
import torch.nn.functional as tf

from aps_amd.libs import ApsRegisters
from aps_amd.nn_ops import linear
from aps_amd.task.base import Task
from aps_amd.task.objf import hybrid_permu_objf
from aps_amd.transform.utils import mel_filter


class SepTask(Task):
    """base class of the separation / enhancement tasks (sse.py:27-58)"""

    def __init__(self, nnet: nn.Module, ctx: Optional[nn.Module] = None, description: str = "",
                 weight: Optional[str] = None) -> None:
        super(SepTask, self).__init__(nnet, ctx=ctx, description=description)
        self.weight = None if weight is None else list(map(float, weight.split(",")))

    def objf(self, out, ref):
        raise NotImplementedError

    def transform(self, tensor):
        raise NotImplementedError


class FreqSaTask(SepTask):
    """frequency-domain spectral approximation (sse.py:207-323): MSA / PSA / tPSA targets from the
    polar STFT of mixture and references, masks applied to the mixture magnitude"""

    def __init__(self, nnet: nn.Module, phase_sensitive: bool = False, truncated: float = -1,
                 permute: bool = True, masking: bool = True, num_spks: int = 2,
                 description: str = "", dpcl_weight: float = 0,
                 weight: Optional[str] = None) -> None:
        sa_ctx = nnet.enh_transform.ctx("forward_stft")
        super(FreqSaTask, self).__init__(nnet, ctx=sa_ctx, weight=weight, description=description)
        if not masking and truncated > 0:
            raise ValueError("Conflict parameters: masksing = True while truncated > 0")
        if dpcl_weight > 0:
            raise NotImplementedError("aps_amd FreqSaTask: the DPCL branch is out of scope")
        self.phase_sensitive = phase_sensitive
        self.truncated = truncated
        self.permute = permute
        self.masking = masking
        self.num_spks = num_spks

    def _ref_mag(self, mix_in_polar: th.Tensor, ref_in_polar: th.Tensor,
                 phase_sensitive: bool = False, truncated: float = -1) -> th.Tensor:
        """reference magnitude for SA (sse.py:261-278)"""
        ref_mag, ref_pha = ref_in_polar[..., 0], ref_in_polar[..., 1]
        if phase_sensitive:
            ref_mag = ref_mag * th.clamp(th.cos(ref_pha - mix_in_polar[..., 1]), min=0)
        if truncated > 0:
            ref_mag = th.min(ref_mag, truncated * mix_in_polar[..., 0])
        return ref_mag

    def forward(self, egs: Dict) -> Dict:
        """egs: mix N x (C) x S, ref N x S or [N x S, ...] -> {"loss": scalar}"""
        mix, ref = egs["mix"], egs["ref"]
        mask = self.nnet(mix)
        with th.no_grad():  # targets: functions of the data only
            mix_in_polar = self.ctx(mix[:, 0] if mix.dim() == 3 else mix, return_polar=True)
            if isinstance(mask, th.Tensor):
                mask, ref = [mask], [ref]
            ref_in_polar = [self.ctx(r, return_polar=True) for r in ref]
            targets = [self._ref_mag(mix_in_polar, r, phase_sensitive=self.phase_sensitive,
                                     truncated=self.truncated) for r in ref_in_polar]
        out = [m * mix_in_polar[..., 0] for m in mask] if self.masking else mask
        loss = hybrid_permu_objf(out, targets, self.objf, transform=self.transform,
                                 weight=self.weight, permute=self.permute,
                                 permu_num_spks=self.num_spks)
        return {"loss": loss.mean()}


@ApsRegisters.task.register("sse@freq_linear_sa")
class LinearFreqSaTask(FreqSaTask):
    """linear spectral approximation, L1 or L2 (sse.py:326-381)"""

    def __init__(self, nnet: nn.Module, phase_sensitive: bool = False, truncated: float = -1,
                 permute: bool = True, masking: bool = True, dpcl_weight: float = 0,
                 num_spks: int = 2, objf: str = "L2", weight: Optional[str] = None) -> None:
        super(LinearFreqSaTask, self).__init__(
            nnet, phase_sensitive=phase_sensitive, truncated=truncated, permute=permute,
            masking=masking, weight=weight, dpcl_weight=dpcl_weight, num_spks=num_spks,
            description="Using spectral approximation (MSA or tPSA) loss function")
        self.objf_ptr = tf.l1_loss if objf == "L1" else tf.mse_loss

    def objf(self, out: th.Tensor, ref: th.Tensor) -> th.Tensor:
        """N x F x T pairs -> N"""
        return th.sum(self.objf_ptr(out, ref, reduction="none").mean(-1), -1)

    def transform(self, tensor: th.Tensor) -> th.Tensor:
        return tensor


@ApsRegisters.task.register("sse@freq_mel_sa")
class MelFreqSaTask(FreqSaTask):
    """mel-spectrogram approximation (sse.py:383-455); the mel projection runs on aps_linear over the
    bin-fastest layout the STFT kernel writes (N x T x F rows), returned as the reference's
    N x M x T view"""

    def __init__(self, nnet: nn.Module, phase_sensitive: bool = False, truncated: float = -1,
                 weight: Optional[str] = None, dpcl_weight: float = 0, permute: bool = True,
                 num_spks: int = 2, masking: bool = True, power_mag: bool = False,
                 num_bins: int = 257, num_mels: int = 80, mel_log: int = False, mel_scale: int = 1,
                 mel_norm: bool = False, sr: int = 16000, fmax: int = 8000) -> None:
        super(MelFreqSaTask, self).__init__(
            nnet, phase_sensitive=phase_sensitive, truncated=truncated, permute=permute,
            masking=masking, weight=weight, dpcl_weight=dpcl_weight, num_spks=num_spks,
            description="Using L2 loss of the mel features")
        mel = mel_filter(None, num_bins=num_bins, sr=sr, num_mels=num_mels, fmax=fmax, norm=mel_norm)
        self.mel = nn.Parameter(mel[..., None] * mel_scale, requires_grad=False)
        self.log = mel_log
        self.power_mag = power_mag

    def transform(self, tensor: th.Tensor) -> th.Tensor:
        """N x F x T -> N x M x T"""
        if self.power_mag:
            tensor = tensor**2
        mel = linear(tensor.transpose(1, 2), self.mel[..., 0]).transpose(1, 2)
        return th.log(1 + mel) if self.log else mel

    def objf(self, out: th.Tensor, ref: th.Tensor) -> th.Tensor:
        return th.sum(tf.mse_loss(out, ref, reduction="none").mean(-1), -1)
