"""
Feature transform for ASR -- the surface of aps/transform/asr.py (FeatureTransform registered as
"asr", i.e. aps.transform.AsrTransform) on the MI355X kernels.

The layer classes, their ctor arguments, `exportable()/dim()` methods, the token grammar of
`feats` and the frozen parameters (`transform.N.K`, `.w`, `.filters`, ...) follow the reference so
that yaml configs and checkpoints load unchanged.  Execution differs: `FeatureTransform.forward`
walks the layer list and fuses every run it recognises into ONE kernel launch
    spectrogram|fbank [-log] [-cmvn]  ->  STFT kernel + row-feature kernel
    abs [-mel] [-log] [-cmvn]         ->  one row-feature kernel on the complex input
    runs of pow / mel / log / cmvn    ->  one row-feature kernel
Layers used stand-alone run the same kernels individually.

The training-time randomised tokens `perturb` (speed perturbation) and `aug` (SpecAugment) build
their layers with the reference's parameters and are the identity in eval mode, like the
reference's; in training mode the random draws are the reference's own (same generators, same
order, so a seeded run reproduces its output) and csrc/augment.hip applies them.
"""
import math
import random
import warnings
from typing import List, Optional, Tuple, Union

import torch as th
import torch.nn as nn

from aps_amd import _native as nat
from aps_amd.const import EPSILON, MAX_INT16
from aps_amd.cplx import ComplexTensor
from aps_amd.libs import ApsRegisters
from aps_amd.nn_ops import linear
from aps_amd.ops import (MelBands, NanGuard, SpectralPlan, abs_features, cmvn_global, length_map,
                         cmvn_utterance, delta, row_features, splice, store_features)
from aps_amd.spectrogram import packed_view, store_of
from aps_amd.transform.utils import STFT, mel_filter, speed_perturb_filter, stft_features

AsrReturnType = Union[th.Tensor, Optional[th.Tensor]]


def check_valid(feature: th.Tensor,
                num_frames: Optional[th.Tensor],
                nan_guard: Optional[NanGuard] = None,
                nan_policy: str = "sync") -> Tuple[th.Tensor]:
    """NaN check + trim to max(num_frames) (asr.py:33-53).  With a NanGuard the NaN scan already
    happened inside the kernel that wrote `feature`; without one the tensor is scanned here."""
    shape = feature.shape
    if nan_guard is not None:
        nan_guard.after_launch(nan_policy, shape, feature.device if feature.is_cuda else None)
    elif nan_policy != "off":
        num_nans = th.sum(th.isnan(feature))
        if num_nans:
            raise ValueError(f"Detect {num_nans} NANs in feature matrices, shape = {shape}...")
    if num_frames is not None:
        max_frames = int(num_frames.max().item())
        if feature.shape[-2] < max_frames:
            raise RuntimeError(f"feats shape: {shape[-2]} x {shape[-1]}, " +
                               f"num_frames = {num_frames.tolist()}")
        if feature.shape[-2] > max_frames:
            feature = feature[..., :max_frames, :]
    return feature, num_frames


class RescaleTransform(nn.Module):
    """Rescale audio samples to the int16 range (asr.py:56-84)"""

    def __init__(self, rescale: float = MAX_INT16 * 1.0) -> None:
        super(RescaleTransform, self).__init__()
        self.rescale = rescale

    def extra_repr(self) -> str:
        return f"rescale={self.rescale}"

    def exportable(self) -> bool:
        return False

    def forward(self, wav: th.Tensor) -> th.Tensor:
        return th.round(wav * self.rescale)


class PreEmphasisTransform(nn.Module):
    """Utterance level pre-emphasis, in place like the reference (asr.py:87-113)"""

    def __init__(self, pre_emphasis: float = 0) -> None:
        super(PreEmphasisTransform, self).__init__()
        self.pre_emphasis = pre_emphasis

    def extra_repr(self) -> str:
        return f"pre_emphasis={self.pre_emphasis}"

    def exportable(self) -> bool:
        return False

    def forward(self, wav: th.Tensor) -> th.Tensor:
        if self.pre_emphasis > 0:
            wav[..., 1:] = wav[..., 1:] - self.pre_emphasis * wav[..., :-1]
        return wav


class TFTransposeTransform(nn.Module):
    """Swap time/frequency axis (a view)"""

    def __init__(self, axis1: int = -1, axis2: int = -2) -> None:
        super(TFTransposeTransform, self).__init__()
        self.axis1 = axis1
        self.axis2 = axis2

    def extra_repr(self) -> str:
        return f"axis1={self.axis1}, axis2={self.axis2}"

    def exportable(self) -> bool:
        return True

    def forward(self, tensor: th.Tensor) -> th.Tensor:
        return tensor.transpose(-1, -2)


class SpectrogramTransform(STFT):
    """STFT layer at the head of spectrogram / fbank chains (asr.py:226-277)"""

    def __init__(self,
                 frame_len: int,
                 frame_hop: int,
                 center: bool = False,
                 window: str = "hamm",
                 round_pow_of_two: bool = True,
                 normalized: bool = False,
                 pre_emphasis: float = 0.97,
                 onesided: bool = True,
                 mode: str = "librosa") -> None:
        super(SpectrogramTransform, self).__init__(frame_len,
                                                   frame_hop,
                                                   center=center,
                                                   window=window,
                                                   round_pow_of_two=round_pow_of_two,
                                                   pre_emphasis=pre_emphasis,
                                                   normalized=normalized,
                                                   onesided=onesided,
                                                   mode=mode)

    def dim(self) -> int:
        return self.num_bins

    def exportable(self) -> bool:
        return False

    def forward(self, wav: th.Tensor) -> th.Tensor:
        """N x (C) x S -> N x (C) x F x T x 2"""
        return super().forward(wav, return_polar=False)


class MagnitudeTransform(nn.Module):
    """[real, imag] -> magnitude over the last axis (asr.py:280-303)"""

    def __init__(self, dim: int = -1, eps: float = 0):
        super(MagnitudeTransform, self).__init__()
        self.dim = dim
        self.eps = eps

    def extra_repr(self) -> str:
        return f"dim={self.dim}, eps={self.eps}"

    def exportable(self) -> bool:
        return True

    def forward(self, inp: th.Tensor) -> th.Tensor:
        """N x (C) x F x T x 2 -> N x (C) x F x T"""
        if self.dim not in (-1, inp.dim() - 1) or self.eps != 0 or inp.dim() not in (4, 5):
            # the layer on its own with another axis / eps: sqrt(sum(inp^2, dim) + eps), element-wise
            from aps_amd.ops import reim_axis
            return reim_axis(inp, self.dim, 1, self.eps)
        mag = _magnitude_rows(store_of(inp), SpectralPlan())  # N x (C) x T x F
        return mag.transpose(-1, -2)


def _magnitude_rows(store: th.Tensor, plan: SpectralPlan, nan_flag=None) -> th.Tensor:
    """store N x (C) x T x F x 2 -> N x (C) x T x D, every channel (channels folded into N)"""
    if store.dim() == 4:
        return store_features(store, plan, 0, nan_flag=nan_flag)
    N, Cn = store.shape[:2]
    if store.stride(0) != Cn * store.stride(1):
        store = store.contiguous()
    flat = store.reshape(N * Cn, *store.shape[2:])
    out = store_features(flat, plan, 0, nan_flag=nan_flag)
    return out.view(N, Cn, *out.shape[1:])


class AbsTransform(nn.Module):
    """|.|; a complex input gets eps added to its real part first (asr.py:306-332)"""

    def __init__(self, eps: float = 1e-6) -> None:
        super(AbsTransform, self).__init__()
        self.eps = eps

    def extra_repr(self) -> str:
        return f"eps={self.eps:.3e}"

    def exportable(self) -> bool:
        return True

    def forward(self, tensor: Union[th.Tensor, ComplexTensor]) -> th.Tensor:
        if isinstance(tensor, th.Tensor):
            return tensor.abs()
        return abs_features(_complex_rows(tensor), SpectralPlan(), self.eps)


def _complex_rows(c: ComplexTensor) -> th.Tensor:
    """ComplexTensor (..., F) -> interleaved (..., F, 2); zero-copy for our own kernel outputs"""
    r, i = c.real, c.imag
    if nat.needs_grad(r, i):
        return th.stack([r, i], -1)  # autograd: a differentiable interleave
    if (r.dtype == th.float32 and r.stride() == i.stride() and r.stride(-1) == 2 and
            r.untyped_storage().data_ptr() == i.untyped_storage().data_ptr() and
            i.storage_offset() == r.storage_offset() + 1):
        return th.as_strided(r, (*r.shape, 2), (*r.stride(), 1), r.storage_offset())
    return th.stack([r, i], -1).float().contiguous()


class PowerTransform(nn.Module):
    """x ** power (asr.py:335-357)"""

    def __init__(self, power: float = 2) -> None:
        super(PowerTransform, self).__init__()
        self.power = power

    def extra_repr(self) -> str:
        return f"power={self.power}"

    def exportable(self) -> bool:
        return True

    def forward(self, tensor: th.Tensor) -> th.Tensor:
        if self.power == 1:
            return tensor
        if self.power != 2:
            raise NotImplementedError("PowerTransform: only power 1 or 2")
        return row_features(tensor, SpectralPlan(power=2))


class MelTransform(nn.Module):
    """Multiply by the mel filterbank; parameter `filters` [num_mels, F] (asr.py:360-428)"""

    def __init__(self,
                 frame_len: int,
                 round_pow_of_two: bool = True,
                 sr: int = 16000,
                 num_mels: int = 80,
                 fmin: float = 0.0,
                 fmax: Optional[float] = None,
                 mel_matrix: str = "",
                 coeff_norm: bool = False,
                 requires_grad: bool = False) -> None:
        super(MelTransform, self).__init__()
        if mel_matrix:
            filters = th.load(mel_matrix)
        else:
            filters = mel_filter(frame_len,
                                 round_pow_of_two=round_pow_of_two,
                                 sr=sr,
                                 num_mels=num_mels,
                                 fmax=fmax,
                                 fmin=fmin,
                                 norm=coeff_norm)
        self.num_mels, self.num_bins = filters.shape
        self.filters = nn.Parameter(filters, requires_grad=requires_grad)
        self.fmin = fmin
        self.fmax = sr // 2 if fmax is None else fmax
        self.init = mel_matrix if mel_matrix else "librosa"

    def dim(self) -> int:
        return self.num_mels

    def exportable(self) -> bool:
        return True

    def extra_repr(self) -> str:
        shape = self.filters.shape
        return (f"fmin={self.fmin}, fmax={self.fmax}, " +
                f"mel_filter={shape[0]}x{shape[1]}, init={self.init}")

    def trainable(self) -> bool:
        """are the filters being trained in this call? (requires_grad=True, asr.py:400: the
        projection then runs as a GEMM under autograd instead of inside a fused feature launch)"""
        return self.filters.requires_grad and th.is_grad_enabled()

    def bands(self) -> MelBands:
        if self.trainable():
            raise RuntimeError("MelTransform: trainable filters take the autograd path (forward()), "
                               "not the banded form of the fused launches")
        return MelBands.cached(self, self.filters)

    def forward(self, linear: th.Tensor) -> th.Tensor:
        """N x (C) x T x F -> N x (C) x T x B"""
        if linear.dim() not in [3, 4]:
            raise RuntimeError("MelTransform expect 3/4D tensor, " +
                               f"but got {linear.dim()} instead")
        if self.trainable() or (th.is_grad_enabled() and linear.requires_grad):
            from aps_amd.nn_ops import linear as gemm  # tf.linear(x, filters), asr.py:427
            return gemm(linear, self.filters)
        return row_features(linear, SpectralPlan(mel=self.bands()))


class LogTransform(nn.Module):
    """log(clamp(x, eps)) or log(lower_bound + x) (asr.py:431-464)"""

    def __init__(self, eps: float = 1e-5, lower_bound: float = 0.0) -> None:
        super(LogTransform, self).__init__()
        self.eps = eps
        self.lower_bound = lower_bound

    def dim_scale(self) -> int:
        return 1

    def exportable(self) -> bool:
        return True

    def extra_repr(self) -> str:
        return f"eps={self.eps:.3e}, lower_bound={self.lower_bound}"

    def forward(self, linear: th.Tensor) -> th.Tensor:
        return row_features(linear, _fuse_tail([self])[0])


class CmvnTransform(nn.Module):
    """Utterance / global mean & variance normalisation (asr.py:520-618).
    NB (kept from the reference): "per_band" statistics run over the LAST axis, i.e. per frame."""

    def __init__(self,
                 norm_mean: bool = True,
                 norm_var: bool = True,
                 per_band: bool = True,
                 dim: int = 1,
                 gcmvn: str = "",
                 eps: float = 1e-5) -> None:
        super(CmvnTransform, self).__init__()
        self.gmean, self.gstd = None, None
        if gcmvn:
            if gcmvn.split(".")[-1] == "ark":
                raise NotImplementedError("gcmvn in Kaldi .ark format needs kaldi_python_io")
            try:
                stats = th.load(gcmvn)
                mean, std = stats[0], stats[1]
            except FileNotFoundError:
                warnings.warn(f"{gcmvn} not found (no impact when " +
                              "will load checkpoint later) ...")
                mean = th.zeros(dim)
                std = th.ones(dim)
            self.gmean = nn.Parameter(mean, requires_grad=False)
            self.gstd = nn.Parameter(std, requires_grad=False)
        self.norm_mean = norm_mean
        self.norm_var = norm_var
        self.per_band = per_band
        self.gcmvn = gcmvn
        self.eps = eps

    def extra_repr(self) -> str:
        return (f"norm_mean={self.norm_mean}, norm_var={self.norm_var}, " +
                f"per_band={self.per_band}, gcmvn_stats={self.gcmvn}, eps={self.eps:.3e}")

    def dim_scale(self) -> int:
        return 1

    def exportable(self) -> bool:
        return True

    def fusible(self) -> bool:
        """row statistics: what the feature kernels compute in the same pass"""
        return self.gmean is None and self.per_band

    def forward(self, feats: th.Tensor) -> th.Tensor:
        if not self.norm_mean and not self.norm_var:
            return feats
        if self.fusible():
            return row_features(feats, _fuse_tail([self])[0])
        if self.gmean is not None:
            return cmvn_global(feats, self.gmean, self.gstd, self.norm_mean, self.norm_var)
        return cmvn_utterance(feats, self.norm_mean, self.norm_var, self.eps)


class DiscreteCosineTransform(nn.Module):
    """log-mel -> cepstra: orthonormal DCT-II (+ liftering) as one GEMM (asr.py:467-517).  The
    `dct` / `cepstral_lifter` parameters keep the reference's names and shapes; the matrix is the
    closed form of scipy's `dct(eye(M), norm="ortho")` the reference builds it from."""

    def __init__(self, num_ceps: int = 13, num_mels: int = 40, lifter: float = 0) -> None:
        super(DiscreteCosineTransform, self).__init__()
        self.lifter = lifter
        self.num_ceps = num_ceps
        n = th.arange(num_mels, dtype=th.float64)
        k = th.arange(num_ceps, dtype=th.float64)[:, None]
        mat = th.cos(math.pi * (2 * n + 1) * k / (2 * num_mels)) * math.sqrt(2.0 / num_mels)
        mat[0] *= math.sqrt(0.5)
        self.dct = nn.Parameter(mat.float(), requires_grad=False)  # num_ceps x num_mels
        if lifter > 0:
            cepstral_lifter = 1 + lifter * 0.5 * th.sin(
                math.pi * th.arange(1, 1 + num_ceps) / lifter)
            self.cepstral_lifter = nn.Parameter(cepstral_lifter, requires_grad=False)
        else:
            self.cepstral_lifter = None

    def dim(self) -> int:
        return self.num_ceps

    def dim_scale(self) -> int:
        return 1

    def exportable(self) -> bool:
        return True

    def extra_repr(self) -> str:
        return "cepstral_lifter={0}, dct={1[0]}x{1[1]}".format(self.lifter, self.dct.shape)

    def forward(self, log_mel: th.Tensor) -> th.Tensor:
        """N x (C) x T x B -> N x (C) x T x P"""
        mat = self.dct
        if self.cepstral_lifter is not None:  # (x D^T) * l == x (l[:, None] * D)^T
            mat = mat * self.cepstral_lifter[:, None]
        return linear(log_mel, mat)


class SpliceTransform(nn.Module):
    """feature splicing (edge frames repeated) + frame subsampling (asr.py:687-728)"""

    def __init__(self, lctx: int = 0, rctx: int = 0, subsampling_factor: int = 1) -> None:
        super(SpliceTransform, self).__init__()
        self.subsampling_factor = subsampling_factor
        self.lctx = max(lctx, 0)
        self.rctx = max(rctx, 0)

    def extra_repr(self) -> str:
        return (f"context=({self.lctx}, {self.rctx}), " +
                f"subsampling_factor={self.subsampling_factor}")

    def dim_scale(self) -> int:
        return 1 + self.rctx + self.lctx

    def exportable(self) -> bool:
        return True

    def forward(self, feats: th.Tensor) -> th.Tensor:
        """N x ... x Ti x F -> N x ... x To x FD"""
        if self.lctx + self.rctx == 0 and self.subsampling_factor == 1:
            return feats
        return splice(feats, self.lctx, self.rctx, self.subsampling_factor)


class DeltaTransform(nn.Module):
    """delta / delta-delta features (asr.py:731-782); `scale` is the reference's frozen parameter"""

    def __init__(self, ctx: int = 2, order: int = 2, delta_as_channel: bool = False) -> None:
        super(DeltaTransform, self).__init__()
        self.ctx = ctx
        self.order = order
        scale = th.arange(-ctx, ctx + 1, dtype=th.float32)
        normalizer = sum(i * i for i in range(-ctx, ctx + 1))
        self.scale = nn.Parameter(scale / normalizer, requires_grad=False)
        self.delta_as_channel = delta_as_channel

    def extra_repr(self) -> str:
        return (f"context={self.ctx}, order={self.order}, " +
                f"delta_as_channel={self.delta_as_channel}")

    def dim_scale(self) -> int:
        return self.order

    def exportable(self) -> bool:
        return True

    def forward(self, feats: th.Tensor) -> th.Tensor:
        """N x (C) x T x F -> N x (C) x T x FD (or N x D x T x F with delta_as_channel)"""
        return delta(feats, self.scale, self.ctx, self.order, self.delta_as_channel)


def draw_tf_bands(batch: int, shape: Tuple[int, int], pm: float = 0.0, ps: float = 0.0,
                  max_bands: int = 30, max_frame: int = 40, num_freq_masks: int = 2,
                  num_time_masks: int = 2) -> Tuple[List[List[Tuple[int, int]]], int, int]:
    """The random draws of tf_mask / random_mask (augment.py:13-83) in the reference's order, as
    (begin, length) bands instead of dense masks: per utterance first the frequency bands, then the
    time bands; every band costs random.randint(1, max - 1) and, unless it is skipped for being
    too long, random.randint(0, L - length - 1).  Returns (bands per utterance, num_freq, num_time)."""
    T, F = shape
    max_bands = min(max_bands, F)
    if ps > 0:
        max_frame = min(max_frame, int(T * ps))
    if pm > 0:
        num_time_masks = min(num_time_masks, int(T * pm))
    drawn = []
    for _ in range(batch):
        bands = []
        for size, limit, count in ((F, max_bands, num_freq_masks), (T, max_frame, num_time_masks)):
            for _ in range(count):
                length = random.randint(1, limit - 1)
                if size - length <= 0:
                    bands.append((0, 0))
                    continue
                bands.append((random.randint(0, size - length - 1), length))
        drawn.append(bands)
    return drawn, num_freq_masks, num_time_masks


class SpeedPerturbTransform(nn.Module):
    """Speed perturbation (asr.py:116-195): a TRAINING-time randomised layer -- identity in eval
    mode, as in the reference.  Holds the reference's frozen resampling filters and rate buffers
    (`weights.N`, `src_sr`, `dst_sr`).  In training mode every utterance draws one of the factors
    (th.randint, as the reference does) and the batch is resampled in one aps_speed_perturb launch."""

    def __init__(self, sr: int = 16000, perturb: str = "0.9,1.0,1.1") -> None:
        super(SpeedPerturbTransform, self).__init__()
        self.sr = sr
        self.factor_str = perturb
        dst_sr = [int(factor * sr) for factor in map(float, perturb.split(","))]
        if not len(dst_sr):
            raise ValueError("No perturb options for doing speed perturb")
        if sr not in dst_sr:
            raise ValueError(f"We should keep 1.0 in perturb options: {perturb}")
        self.weights = nn.ParameterList([
            nn.Parameter(speed_perturb_filter(sr, fs), requires_grad=False)
            for fs in dst_sr if fs != sr
        ])
        shapes = [w.shape for w in self.weights]
        self.register_buffer("src_sr", th.tensor([s[1] for s in shapes] + [1], dtype=th.int64))
        self.register_buffer("dst_sr", th.tensor([s[0] for s in shapes] + [1], dtype=th.int64))
        self.last_choice = None

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(sr={self.sr}, factor={self.factor_str})"

    def exportable(self) -> bool:
        return False

    def output_length(self, inp_len: Optional[th.Tensor]) -> Optional[th.Tensor]:
        """lengths after the last forward's resampling (asr.py:153-164)"""
        if self.last_choice is None:
            return inp_len
        if inp_len is None:
            return None
        choice = self.last_choice.to(self.src_sr.device)
        return th.div(inp_len, self.src_sr[choice].to(inp_len.device),
                      rounding_mode="trunc") * self.dst_sr[choice].to(inp_len.device)

    def forward(self, wav: th.Tensor) -> th.Tensor:
        """N x S -> N x S' (training; S' = the longest resampled utterance), identity in eval"""
        self.last_choice = None
        if not self.training:
            return wav
        if wav.dim() != 2:
            raise RuntimeError(f"Now only supports 2D tensor, got {wav.dim()}")
        import ctypes as C
        from aps_amd import _native as nat
        nat.require_device(wav, *self.weights)
        lib = nat.load()
        N, S = wav.shape
        K = len(self.weights)
        choice = th.randint(0, K + 1, (N,))  # the reference's draw, on the host generator
        self.last_choice = choice
        banks = [nat.f32c(w) for w in self.weights]
        lengths = [S if c == K else (S // banks[c].shape[1]) * banks[c].shape[0]
                   for c in choice.tolist()]
        for c in set(choice.tolist()):
            if c != K and S // banks[c].shape[1] == 0:
                raise RuntimeError(f"Input wav is too short to be perturbed, length = {S}")
        S_out = max(lengths)
        out = th.empty(N, S_out, device=wav.device, dtype=th.float32)
        ptrs = (C.c_void_p * max(K, 1))(*[b.data_ptr() for b in banks])
        src = (C.c_int32 * max(K, 1))(*[b.shape[1] for b in banks])
        dst = (C.c_int32 * max(K, 1))(*[b.shape[0] for b in banks])
        taps = (C.c_int32 * max(K, 1))(*[b.shape[2] for b in banks])
        rc = lib.aps_speed_perturb(nat.ptr(nat.f32c(wav)), nat.ptr(choice.to(wav.device)), ptrs, src,
                                   dst, taps, K, nat.ptr(out), N, S, S_out, nat.stream_of(wav))
        nat.check(rc, "aps_speed_perturb")
        return out


class SpecAugTransform(nn.Module):
    """SpecAugment (asr.py:621-684): a TRAINING-time randomised layer -- identity in eval mode, as
    in the reference.  In training mode the coin flip (th.rand) and the band draws
    (random.randint, `draw_tf_bands`) are the reference's own, in its order; aps_spec_augment
    applies the bands (zeros or the mean of the input)."""

    def __init__(self, p: float = 0.5, adaptive_args: Tuple[float] = (0.0, 0.0),
                 time_args: Tuple[int] = (40, 1), freq_args: Tuple[int] = (30, 1),
                 mask_zero: bool = True) -> None:
        super(SpecAugTransform, self).__init__()
        assert len(freq_args) == 2 and len(time_args) == 2
        self.fnum, self.tnum = freq_args[1], time_args[1]
        self.mask_zero = mask_zero
        self.F, self.T = freq_args[0], time_args[0]
        self.p = p
        self.pm, self.ps = adaptive_args

    def extra_repr(self) -> str:
        return (f"max_bands={self.F}, max_frame={self.T}, p={self.p}, pm={self.pm}, ps={self.ps}, "
                f"mask_zero={self.mask_zero}, num_freq_masks={self.fnum}, "
                f"num_time_masks={self.tnum}")

    def exportable(self) -> bool:
        return False

    def forward(self, x: th.Tensor) -> th.Tensor:
        """N x (C) x T x F -> same shape"""
        if not (self.training and th.rand(1).item() < self.p):
            return x
        if x.dim() not in (3, 4):
            raise RuntimeError(f"SpecAugTransform expects 3/4D tensor, got {x.dim()}D")
        from aps_amd import _native as nat
        nat.require_device(x)
        lib = nat.load()
        N, T, F = x.shape[0], x.shape[-2], x.shape[-1]
        Cn = x.shape[1] if x.dim() == 4 else 1
        drawn, nf, nt = draw_tf_bands(N, (T, F), pm=self.pm, ps=self.ps, max_bands=self.F,
                                      max_frame=self.T, num_freq_masks=self.fnum,
                                      num_time_masks=self.tnum)
        if nf + nt == 0:
            return x
        bands = th.tensor(drawn, dtype=th.int32).reshape(N, nf + nt, 2).to(x.device)
        xc = nat.f32c(x)
        out = th.empty_like(xc)
        work = None if self.mask_zero else th.empty(1, device=x.device, dtype=th.float64)
        rc = lib.aps_spec_augment(nat.ptr(xc), nat.ptr(bands), nat.ptr(out), N, Cn, T, F, nf, nt,
                                  int(self.mask_zero), nat.ptr(work), nat.stream_of(x))
        nat.check(rc, "aps_spec_augment")
        return out


def _fuse_tail(layers: List[nn.Module], plan: Optional[SpectralPlan] = None):
    """Greedily absorb [Power] [Mel] [Log] [Cmvn(row)] (in that order) into `plan`.
    Returns (plan, number of layers consumed)."""
    plan = plan if plan is not None else SpectralPlan()
    used = 0
    stage = 0  # 0: power, 1: mel, 2: log, 3: cmvn
    for layer in layers:
        if isinstance(layer, PowerTransform) and stage <= 0 and layer.power in (1, 2):
            plan.power = int(layer.power)
            stage = 1
        elif isinstance(layer, MelTransform) and stage <= 1:
            if layer.trainable():  # its own GEMM under autograd: the fused run ends in front of it
                break
            plan.mel = layer.bands()
            stage = 2
        elif isinstance(layer, LogTransform) and stage <= 2:
            plan.apply_log = True
            plan.log_eps = layer.eps
            plan.log_lower_bound = layer.lower_bound
            stage = 3
        elif isinstance(layer, CmvnTransform) and stage <= 3 and layer.fusible():
            plan.norm_mean = layer.norm_mean
            plan.norm_var = layer.norm_var
            plan.cmvn_eps = layer.eps
            used += 1
            break
        else:
            break
        used += 1
    return plan, used


@ApsRegisters.transform.register("asr")
class FeatureTransform(nn.Module):
    """
    Feature transform for ASR tasks (asr.py:784-1033): same 38 ctor kwargs and defaults, same
    attributes (`transform`, `spectra_index`, `perturb_index`, `feats_dim`, `subsampling_factor`)
    and the same forward contract: (inp_pad N x (C) x S, inp_len N | None) -> (feats, num_frames).

    Extra attribute (not a ctor argument, so yaml dicts stay valid): `nan_policy` in
    {"sync", "deferred", "off"}, see aps_amd.ops.NanGuard.  Default "sync" = reference behaviour.
    """

    def __init__(self,
                 feats: str = "fbank-log-cmvn",
                 frame_len: int = 400,
                 frame_hop: int = 160,
                 window: str = "hamm",
                 center: bool = False,
                 round_pow_of_two: bool = True,
                 stft_normalized: bool = False,
                 stft_mode: str = "librosa",
                 audio_norm: bool = True,
                 pre_emphasis: float = 0.97,
                 use_power: bool = False,
                 sr: int = 16000,
                 speed_perturb: str = "0.9,1.0,1.1",
                 log_lower_bound: float = 0,
                 num_mels: int = 80,
                 mel_matrix: str = "",
                 mel_coeff_norm: bool = False,
                 min_freq: int = 0,
                 max_freq: Optional[int] = None,
                 num_ceps: int = 13,
                 lifter: float = 0,
                 aug_prob: float = 0,
                 aug_adaptive_args: Tuple[float] = (0, 0),
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
                 eps: float = EPSILON) -> None:
        super(FeatureTransform, self).__init__()
        if not feats:
            raise ValueError("FeatureTransform: \'feats\' can not be empty")
        feat_tokens = feats.split("-")
        transform = [] if audio_norm else [RescaleTransform()]
        feats_dim = 0
        stft_kwargs = {
            "mode": stft_mode,
            "window": window,
            "center": center,
            "normalized": stft_normalized,
            "pre_emphasis": pre_emphasis,
            "round_pow_of_two": round_pow_of_two
        }
        mel_kwargs = {
            "round_pow_of_two": round_pow_of_two,
            "sr": sr,
            "fmin": min_freq,
            "fmax": max_freq,
            "num_mels": num_mels,
            "coeff_norm": mel_coeff_norm,
            "mel_matrix": mel_matrix,
            "requires_grad": requires_grad
        }
        self.spectra_index = -1
        self.perturb_index = -1
        for tok in feat_tokens:
            if tok == "perturb":
                self.perturb_index = len(transform)
                transform.append(SpeedPerturbTransform(sr=sr, perturb=speed_perturb))
            elif tok == "aug":
                transform.append(SpecAugTransform(p=aug_prob, adaptive_args=aug_adaptive_args,
                                                  mask_zero=aug_mask_zero, time_args=aug_time_args,
                                                  freq_args=aug_freq_args))
            elif tok == "emph":
                transform.append(PreEmphasisTransform(pre_emphasis=pre_emphasis))
            elif tok in ("spectrogram", "fbank", "mfcc"):
                self.spectra_index = len(transform)
                transform += [
                    SpectrogramTransform(frame_len, frame_hop, **stft_kwargs),
                    MagnitudeTransform(dim=-1),
                    TFTransposeTransform(),
                    PowerTransform(power=2 if use_power else 1)
                ]
                feats_dim = transform[self.spectra_index].dim()
                if tok in ("fbank", "mfcc"):
                    transform.append(MelTransform(frame_len, **mel_kwargs))
                    feats_dim = transform[-1].dim()
                if tok == "mfcc":
                    transform += [LogTransform(eps=eps, lower_bound=log_lower_bound),
                                  DiscreteCosineTransform(num_ceps=num_ceps, num_mels=num_mels,
                                                          lifter=lifter)]
                    feats_dim = transform[-1].dim()
            elif tok == "trans":
                transform.append(TFTransposeTransform())
            elif tok == "pow":
                transform.append(PowerTransform())
            elif tok == "mel":
                transform.append(MelTransform(frame_len, **mel_kwargs))
                feats_dim = transform[-1].dim()
            elif tok == "log":
                transform.append(LogTransform(eps=eps, lower_bound=log_lower_bound))
            elif tok == "abs":
                transform.append(AbsTransform(eps=eps))
            elif tok == "cmvn":
                transform.append(
                    CmvnTransform(norm_mean=norm_mean,
                                  norm_var=norm_var,
                                  per_band=norm_per_band,
                                  gcmvn=gcmvn,
                                  dim=feats_dim,
                                  eps=eps))
            elif tok == "dct":
                transform.append(DiscreteCosineTransform(num_ceps=num_ceps, num_mels=num_mels,
                                                         lifter=lifter))
                feats_dim = transform[-1].dim()
            elif tok == "splice":
                transform.append(SpliceTransform(lctx=lctx, rctx=rctx,
                                                 subsampling_factor=subsampling_factor))
                feats_dim *= (1 + lctx + rctx)
            elif tok == "delta":
                transform.append(DeltaTransform(ctx=delta_ctx, order=delta_order,
                                                delta_as_channel=delta_as_channel))
                feats_dim *= (1 + delta_order)
            else:
                raise RuntimeError(f"Unknown token {tok} in {feats}")
        self.transform = nn.Sequential(*transform)
        self.feats_dim = feats_dim
        self.subsampling_factor = subsampling_factor
        self.nan_policy = "sync"
        self._nan_guard = NanGuard()

    def dim(self) -> int:
        return self.feats_dim

    def num_frames(self, inp_len: Optional[th.Tensor]) -> Optional[th.Tensor]:
        """number of frames per utterance (asr.py:1003-1019)"""
        if inp_len is None:
            return None
        if self.spectra_index == -1:
            warnings.warn("SpectrogramTransform layer is not found, " +
                          "return input as the #num_frames")
            return inp_len
        if self.perturb_index != -1:
            inp_len = self.transform[self.perturb_index].output_length(inp_len)
        num_frames = self.transform[self.spectra_index].num_frames(inp_len)
        if self.subsampling_factor == 1:
            return num_frames
        return length_map(num_frames, 0, self.subsampling_factor, 0)

    def _run(self, x, nan_flag):
        """walk the layer list, fusing recognised runs into single launches"""
        layers = list(self.transform)
        i = 0
        while i < len(layers):
            layer = layers[i]
            if (isinstance(layer, SpectrogramTransform) and i + 2 < len(layers) and
                    isinstance(layers[i + 1], MagnitudeTransform) and
                    isinstance(layers[i + 2], TFTransposeTransform) and
                    layers[i + 1].eps == 0):
                plan, used = _fuse_tail(layers[i + 3:])
                fused = None
                if x.dim() in (2, 3) and x.is_cuda:
                    # STFT -> |X| -> ... in ONE launch, the spectrogram is never materialised
                    lead = x.shape[:-1]
                    fused = stft_features(x.reshape(-1, 1, x.shape[-1]), layer._kernel_window(),
                                          layer.fft_size, layer.frame_hop, plan, 0, None, False,
                                          center=layer.center, pre_emphasis=layer.pre_emphasis,
                                          normalized=layer.normalized, write_store=False,
                                          nan_flag=nan_flag) if layer.onesided else None
                if fused is not None:
                    x = fused[1].view(*lead, *fused[1].shape[1:])
                else:
                    x = _magnitude_rows(layer.to_store(x), plan, nan_flag)
                i += 3 + used
            elif isinstance(layer, AbsTransform) and isinstance(x, ComplexTensor):
                plan, used = _fuse_tail(layers[i + 1:])
                x = abs_features(_complex_rows(x), plan, layer.eps, nan_flag)
                i += 1 + used
            elif isinstance(layer, (PowerTransform, MelTransform, LogTransform, CmvnTransform)):
                plan, used = _fuse_tail(layers[i:])
                if used == 0:
                    x = layer(x)
                    used = 1
                else:
                    x = row_features(x, plan, nan_flag)
                i += used
            else:
                x = layer(x)
                i += 1
        return x

    def abs_chain(self):
        """(plan, eps) when this transform, applied to a COMPLEX input, is exactly AbsTransform + a tail one
        row kernel takes ([pow] [mel] [log] [row cmvn]) -- `abs-mel-log-cmvn`, what EnhASRBase applies to the
        beamformer's output (enh_att.py:92-93) -- so that a producer may compute it on its way out
        (mvdr.beamform_features); None otherwise"""
        layers = list(self.transform)
        if not layers or not isinstance(layers[0], AbsTransform):
            return None
        plan, used = _fuse_tail(layers[1:])
        return (plan, layers[0].eps) if 1 + used == len(layers) else None

    def finish(self, feats: th.Tensor, inp_len: Optional[th.Tensor]) -> AsrReturnType:
        """the end of `forward` for features a producer computed with this transform's own plan and NaN
        counter (`nan_pointer`): frame counts and check_valid"""
        guard = self._nan_guard if self.nan_policy != "off" else None
        return check_valid(feats, self.num_frames(inp_len), guard, self.nan_policy)

    def nan_pointer(self, device):
        guard = self._nan_guard if self.nan_policy != "off" else None
        return guard.pointer(device) if (guard is not None and device.type == "cuda") else None

    def forward(self, inp_pad: Union[th.Tensor, ComplexTensor],
                inp_len: Optional[th.Tensor]) -> AsrReturnType:
        """(N x (C) x S | features, N | None) -> (N x (C) x T x D, num_frames)"""
        dev = inp_pad.device
        guard = self._nan_guard if self.nan_policy != "off" else None
        flag = guard.pointer(dev) if (guard is not None and dev.type == "cuda") else None
        feats = self._run(inp_pad, flag)
        num_frames = self.num_frames(inp_len)
        return check_valid(feats, num_frames, guard, self.nan_policy)
