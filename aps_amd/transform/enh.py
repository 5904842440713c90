"""
Spatial feature transform for enhancement / separation front-ends -- the surface of
aps/transform/enh.py (FeatureTransform registered as "enh", i.e. aps.transform.EnhTransform).

`encode` runs the STFT kernel once for all channels into a bin-fastest store and hands back the
reference-shaped view; `forward` computes log-magnitude/CMVN of the reference channel AND the
cos/sin IPDs of every channel pair in ONE kernel launch over that store (the reference chains
~10 torch ops with several transposing copies, enh.py:595-613); `decode` is the iSTFT kernel.

DfTransform and FixedBeamformer (enh.py:146-384) live in aps_amd/transform/spatial.py.
"""
from typing import List, Optional, Tuple

import torch as th
import torch.nn as nn

from aps_amd.const import EPSILON
from aps_amd.libs import ApsRegisters
from aps_amd.ops import NanGuard, SpectralPlan, store_features
from aps_amd.spectrogram import packed_view, store_of
from aps_amd.transform.asr import (AsrReturnType, MagnitudeTransform, TFTransposeTransform,
                                   check_valid, _fuse_tail)
from aps_amd.transform.asr import FeatureTransform as AsrTransform
from aps_amd.transform.utils import STFT, iSTFT, stft_features


def _split_index(sstr: str) -> Tuple[List[int], List[int]]:
    pair = [tuple(map(int, p.split(","))) for p in sstr.split(";")]
    return [t[0] for t in pair], [t[1] for t in pair]


class RefChannelTransform(nn.Module):
    """Choose one reference channel (enh.py:21-49)"""

    def __init__(self, ref_channel: int = 0, input_dim: int = 4) -> None:
        super(RefChannelTransform, self).__init__()
        self.ref_channel = ref_channel
        self.input_dim = input_dim

    def extra_repr(self) -> str:
        return f"ref_channel={self.ref_channel}"

    def exportable(self) -> bool:
        return True

    def forward(self, inp: th.Tensor) -> th.Tensor:
        if inp.dim() != self.input_dim or self.ref_channel < 0:
            return inp
        return inp[:, self.ref_channel]


class PhaseTransform(nn.Module):
    """[real, imag] -> phase (enh.py:52-76)"""

    def __init__(self, dim: int = -1):
        super(PhaseTransform, self).__init__()
        self.dim = dim

    def extra_repr(self) -> str:
        return f"dim={self.dim}"

    def exportable(self) -> bool:
        return True

    def forward(self, inp: th.Tensor) -> th.Tensor:
        """N x ... x 2 x ... -> N x ...: atan2(imag, real) along `dim` (inside EnhTransform.forward the
        phase is never formed -- the feature kernel works on unit vectors; this is the layer on its own)"""
        from aps_amd.ops import reim_axis
        return reim_axis(inp, self.dim, 0)


class IpdTransform(nn.Module):
    """Inter-channel phase differences (enh.py:79-143); evaluated by the fused feature kernel"""

    def __init__(self, ipd_index: str = "1,0", cos: bool = True, sin: bool = False) -> None:
        super(IpdTransform, self).__init__()
        self.index_l, self.index_r = _split_index(ipd_index)
        self.ipd_index = ipd_index
        self.cos = cos
        self.sin = sin
        self.num_pairs = len(self.index_l) * 2 if cos and sin else len(self.index_l)

    def extra_repr(self) -> str:
        return f"ipd_index={self.ipd_index}, cos={self.cos}, sin={self.sin}"

    def exportable(self) -> bool:
        return True

    def forward(self, p: th.Tensor) -> th.Tensor:
        """phase N x C x T x F (or C x T x F) -> IPD features N x T x MF (enh.py:113-143)"""
        from aps_amd import _native as nat
        from aps_amd.ops import _pair_tensors
        if p.dim() not in [3, 4]:
            raise RuntimeError(f"{self.__class__.__name__} expect 3/4D tensor, but got {p.dim():d} instead")
        if p.dim() == 3:
            p = p.unsqueeze(0)
        N, C, T, F = p.shape
        assert C != 1
        if not self.cos:
            # the reference raises NameError here (enh.py:138-141: `ipd` used before assignment)
            raise NameError("name 'ipd' is not defined (IpdTransform(cos=False), as the reference)")
        if max(self.index_l + self.index_r) >= C or min(self.index_l + self.index_r) < -C:
            raise IndexError(f"IPD pair index out of range for {C} channels: {self.ipd_index}")
        nat.require_device(p)
        il, ir = _pair_tensors(tuple(i % C for i in self.index_l), tuple(i % C for i in self.index_r),
                               p.device)
        P = len(self.index_l)
        out = th.empty(N, T, (2 if self.sin else 1) * P * F, device=p.device, dtype=th.float32)
        rc = nat.load().aps_ipd_from_phase(nat.ptr(nat.f32c(p)), nat.ptr(il), nat.ptr(ir), N, C, T, F, P,
                                           int(self.sin), nat.ptr(out), nat.stream_of(p))
        nat.check(rc, "aps_ipd_from_phase")
        return out


class _IpdChain(nn.Sequential):
    """Phase -> TFTranspose -> Ipd as one launch: packed N x C x F x T x 2 -> N x T x PF"""

    def forward(self, packed: th.Tensor) -> th.Tensor:
        ipd = self[2]
        if not ipd.cos:
            # the reference raises NameError here (enh.py:138-141: `ipd` used before assignment)
            raise NameError("name 'ipd' is not defined (IpdTransform(cos=False), as the reference)")
        if packed.dim() == 4:
            packed = packed[None]
        return store_features(store_of(packed), None, pairs=(ipd.index_l, ipd.index_r),
                              ipd_sin=ipd.sin)


@ApsRegisters.transform.register("enh")
class FeatureTransform(nn.Module):
    """
    Feature transform for enhancement/separation tasks (enh.py:387-613): same ctor kwargs and
    defaults, `encode/decode/forward/ctx/num_frames/dim`, attributes `forward_stft`,
    `inverse_stft`, `mag_transform`, `ipd_transform`, `feats_dim`.
    """

    def __init__(self,
                 feats: str = "spectrogram-log-cmvn",
                 frame_len: int = 512,
                 frame_hop: int = 256,
                 window: str = "sqrthann",
                 round_pow_of_two: bool = True,
                 stft_normalized: bool = False,
                 stft_mode: str = "librosa",
                 center: bool = False,
                 ref_channel: int = 0,
                 use_power: bool = False,
                 sr: int = 16000,
                 log_lower_bound: float = 0,
                 num_mels: int = 80,
                 mel_matrix: str = "",
                 mel_coeff_norm: bool = False,
                 min_freq: int = 0,
                 max_freq: Optional[int] = None,
                 num_ceps: int = 13,
                 lifter: float = 0,
                 aug_prob: float = 0,
                 aug_adaptive_args: Tuple[int] = (0, 0),
                 aug_mask_zero: bool = True,
                 aug_time_args: Tuple[int] = (40, 1),
                 aug_freq_args: Tuple[int] = (30, 1),
                 norm_mean: bool = True,
                 norm_var: bool = True,
                 norm_per_band: bool = True,
                 gcmvn: str = "",
                 subsampling_factor: int = 1,
                 lctx: int = 1,
                 rctx: int = 1,
                 delta_ctx: int = 2,
                 delta_order: int = 2,
                 delta_as_channel: bool = False,
                 requires_grad: bool = False,
                 ipd_index: str = "",
                 cos_ipd: bool = True,
                 sin_ipd: bool = False,
                 eps: float = EPSILON) -> None:
        super(FeatureTransform, self).__init__()
        self.frame_len = frame_len
        self.frame_hop = frame_hop
        self.stft_kwargs = {
            "mode": stft_mode,
            "window": window,
            "center": center,
            "normalized": stft_normalized,
            "round_pow_of_two": round_pow_of_two
        }
        self.forward_stft = self.ctx(name="forward_stft")
        self.inverse_stft = self.ctx(name="inverse_stft")

        feats_dim = 0
        feats_tok = feats.split("-")
        feats_mag = "-".join([t for t in feats_tok if t != "ipd"])
        if feats_mag:
            asr_transform = AsrTransform(feats=feats_mag,
                                         frame_len=frame_len,
                                         frame_hop=frame_hop,
                                         window=window,
                                         round_pow_of_two=round_pow_of_two,
                                         stft_normalized=stft_normalized,
                                         stft_mode=stft_mode,
                                         center=center,
                                         use_power=use_power,
                                         sr=sr,
                                         log_lower_bound=log_lower_bound,
                                         num_mels=num_mels,
                                         mel_matrix=mel_matrix,
                                         mel_coeff_norm=mel_coeff_norm,
                                         min_freq=min_freq,
                                         max_freq=max_freq,
                                         num_ceps=num_ceps,
                                         lifter=lifter,
                                         aug_prob=aug_prob,
                                         aug_adaptive_args=aug_adaptive_args,
                                         aug_mask_zero=aug_mask_zero,
                                         aug_time_args=aug_time_args,
                                         aug_freq_args=aug_freq_args,
                                         norm_mean=norm_mean,
                                         norm_var=norm_var,
                                         norm_per_band=norm_per_band,
                                         gcmvn=gcmvn,
                                         subsampling_factor=subsampling_factor,
                                         lctx=lctx,
                                         rctx=rctx,
                                         delta_ctx=delta_ctx,
                                         delta_order=delta_order,
                                         delta_as_channel=delta_as_channel,
                                         requires_grad=requires_grad)
            if asr_transform.spectra_index == -1:
                raise RuntimeError("Now only support spectrogram/mfcc/fbank features")
            feats_dim = asr_transform.dim()
            # SpectrogramTransform() is replaced by the reference channel selection
            mag_transform = [
                RefChannelTransform(ref_channel=ref_channel, input_dim=5),
            ] + list(asr_transform.transform[1:])
            self.mag_transform = nn.Sequential(*mag_transform)
        else:
            self.mag_transform = None

        feats_spa = "-".join([t for t in feats_tok if t == "ipd"])
        if feats_spa and ipd_index:
            self.ipd_transform = _IpdChain(
                PhaseTransform(dim=-1), TFTransposeTransform(),
                IpdTransform(ipd_index=ipd_index, cos=cos_ipd, sin=sin_ipd))
            num_index = len(ipd_index.split(";"))
            if cos_ipd and sin_ipd:
                feats_dim += num_index * 2 * self.forward_stft.num_bins
            else:
                feats_dim += num_index * self.forward_stft.num_bins
        else:
            self.ipd_transform = None
        self.feats_dim = feats_dim
        self.nan_policy = "sync"
        self._nan_guard = NanGuard()
        # encode() can compute the features in the same launch as the STFT (the
        # spectrogram is written once and never re-read) and forward(packed) hands them out if `packed`
        # is that very tensor.  None = automatic: on for 2 .. 4 channels without a mel layer, where the
        # launch is stft512_frame_feat_kernel (round 4: a wavefront owns one frame of every channel,
        # no exchange between wavefronts); the other shapes would take the per-channel-wavefront form,
        # which measured 57 us against 27 + 25 for the two launches at N = 32 (round 1), and stay
        # two launches.  True / False force it (A/B runs).
        self.fuse_encode_features = None
        self._fused = None

    def dim(self) -> int:
        return self.feats_dim

    def ctx(self, name: str = "forward_stft") -> nn.Module:
        """A fresh STFT / iSTFT layer for tasks (enh.py:553-560)"""
        ctx = {"forward_stft": STFT, "inverse_stft": iSTFT}
        if name not in ctx:
            raise ValueError(f"Unknown task context: {name}")
        return ctx[name](self.frame_len, self.frame_hop, **self.stft_kwargs)

    def num_frames(self, wav_len: Optional[th.Tensor]) -> Optional[th.Tensor]:
        if wav_len is None:
            return None
        return self.forward_stft.num_frames(wav_len)

    def encode(self, wav_pad: th.Tensor, wav_len: Optional[th.Tensor]) -> AsrReturnType:
        """N x (C) x S -> (packed N x (C) x F x T x 2, num_frames)"""
        self._fused = None
        fuse = self.fuse_encode_features
        if fuse is None:
            fuse = wav_pad.dim() == 3 and 2 <= wav_pad.shape[1] <= 4 and \
                not (th.is_grad_enabled() and wav_pad.requires_grad)
        fused = self._encode_fused(wav_pad, auto=self.fuse_encode_features is None) if fuse else None
        if fused is not None:
            store, feats = fused
            packed = packed_view(store)
            self._fused = (store.data_ptr(), store._version, tuple(store.shape), feats)
        else:
            packed = self.forward_stft(wav_pad, return_polar=False)
        return packed, self.num_frames(wav_len)

    def _encode_fused(self, wav_pad: th.Tensor, auto: bool = False):
        """STFT + features in one launch when the configuration allows it, else None"""
        stft = self.forward_stft
        if wav_pad.dim() != 3 or not wav_pad.is_cuda or stft.fft_size != 512 or not stft.onesided:
            return None
        if auto and stft.pre_emphasis > 0:
            return None
        plan, ref, rest = None, 0, []
        try:
            if self.mag_transform is not None:
                plan, ref_layer, rest = self._mag_plan()
                ref = ref_layer.ref_channel
        except NotImplementedError:
            return None
        if rest or (plan is not None and ref < 0):
            return None
        if auto and plan is not None and plan.mel is not None:
            return None  # (the frame-major kernel has no mel stage)
        pairs, use_sin = None, False
        if self.ipd_transform is not None:
            ipd = self.ipd_transform[2]
            if not ipd.cos:
                return None
            pairs, use_sin = (ipd.index_l, ipd.index_r), ipd.sin
        if plan is None and pairs is None:
            return None
        guard = self._nan_guard if self.nan_policy != "off" else None
        flag = guard.pointer(wav_pad.device) if guard is not None else None
        return stft_features(wav_pad, stft._kernel_window(), stft.fft_size, stft.frame_hop, plan,
                             ref, pairs, use_sin, center=stft.center,
                             pre_emphasis=stft.pre_emphasis, normalized=stft.normalized,
                             write_store=True, nan_flag=flag)

    def decode(self, packed: List[th.Tensor]) -> List[th.Tensor]:
        """[N x F x T x 2, ...] -> [N x S, ...]"""
        return [self.inverse_stft(p, return_polar=False) for p in packed]

    def _mag_plan(self):
        """(SpectralPlan, ref_channel, leftover layers) when the magnitude chain has the shape
        RefChannel, Magnitude, TFTranspose, [Power] [Mel] [Log] [Cmvn] ..."""
        layers = list(self.mag_transform)
        if (len(layers) >= 3 and isinstance(layers[0], RefChannelTransform) and
                isinstance(layers[1], MagnitudeTransform) and layers[1].eps == 0 and
                isinstance(layers[2], TFTransposeTransform)):
            plan, used = _fuse_tail(layers[3:])
            return plan, layers[0], layers[3 + used:]
        raise NotImplementedError("EnhTransform: unsupported magnitude chain")

    def forward(self, packed: th.Tensor) -> th.Tensor:
        """packed N x (C) x F x T x 2 -> N x T x D (spectral ++ spatial), one launch"""
        if packed.dim() not in (4, 5):
            raise RuntimeError(f"EnhTransform expects 4/5D STFT, got {packed.dim()}D")
        store = store_of(packed)
        fused, self._fused = self._fused, None
        if (fused is not None and fused[0] == store.data_ptr() and fused[1] == store._version and
                fused[2] == tuple(store.shape)):
            # computed by encode() in the STFT launch; NaN scan happened in that kernel
            guard = self._nan_guard if self.nan_policy != "off" else None
            return check_valid(fused[3], None, guard, self.nan_policy)[0]
        guard = self._nan_guard if self.nan_policy != "off" else None
        flag = guard.pointer(store.device) if (guard is not None and store.is_cuda) else None
        plan, ref, rest = None, 0, []
        if self.mag_transform is not None:
            plan, ref_layer, rest = self._mag_plan()
            ref = ref_layer.ref_channel if store.dim() == 5 else 0
            if store.dim() == 5 and ref < 0:
                # RefChannelTransform(ref_channel < 0) hands on every channel (enh.py:46): the magnitude
                # chain then yields N x C x T x D, which the reference can only return on its own (its
                # th.cat with the 3-D IPD block fails)
                from aps_amd.transform.asr import _magnitude_rows
                mag = _magnitude_rows(store, plan, nan_flag=flag)
                for layer in rest:
                    mag = layer(mag)
                parts = [mag]
                if self.ipd_transform is not None:
                    parts.append(self.ipd_transform(packed))
                return check_valid(th.cat(parts, -1), None, guard, self.nan_policy)[0]
        pairs, use_sin = None, False
        if self.ipd_transform is not None:
            ipd = self.ipd_transform[2]
            if not ipd.cos:
                raise NameError("name 'ipd' is not defined (IpdTransform(cos=False))")
            if store.dim() == 4:
                raise AssertionError("IPD features need a multi-channel STFT")
            pairs, use_sin = (ipd.index_l, ipd.index_r), ipd.sin
        if plan is None and pairs is None:
            raise RuntimeError("EnhTransform: no feature configured")
        if not rest:
            feats = store_features(store, plan, ref, pairs, use_sin, nan_flag=flag)
        else:
            # magnitude chain has layers the fused kernel does not know: run them separately
            parts = []
            mag = store_features(store, plan, ref, nan_flag=flag)
            for layer in rest:
                mag = layer(mag)
            parts.append(mag)
            if pairs is not None:
                parts.append(store_features(store, None, 0, pairs, use_sin, nan_flag=flag))
            feats = th.cat(parts, -1)
        return check_valid(feats, None, guard, self.nan_policy)[0]
