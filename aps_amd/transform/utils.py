"""
STFT / iSTFT layers and their helpers -- the surface of aps/transform/utils.py, backed by the HIP
kernels in aps_amd/csrc/stft.hip.

Kept from the reference (file:line = aps/transform/utils.py):
  init_window :30-59, init_kernel :62-112 (only to materialise the frozen `K`, `w` parameters the
  reference keeps in checkpoints), mel_filter :115-156, forward_stft :472-532,
  inverse_stft :535-591, STFTBase/STFT/iSTFT :594-758 with parameter names `K` [2W,1,L], `w` [L].
Different by design: the transform itself is an FFT on the GPU (not a dense-DFT conv1d) and the
result is a bin-fastest store exposed through a reference-shaped view (aps_amd/spectrogram.py).
Documented deviations: mode="torch" returns the correctly shaped N x C x F x T x 2 tensor for 3-D
input (the reference mis-shapes it, utils.py:405-407); num_frames() does not mutate its argument
(utils.py:658-659 adds win_length in place).
"""
import ctypes as C
import math
from typing import Optional, Tuple

import numpy as np
import torch as th
import torch.nn as nn
import torch.nn.functional as tf

from aps_amd import _native as nat
from aps_amd.const import EPSILON
from aps_amd.spectrogram import alloc_store, packed_view, store_of

_WINDOWS = ["bartlett", "hann", "hamm", "blackman", "rect", "sqrthann"]


def export_jit(transform: nn.Module) -> nn.Module:
    return nn.Sequential(*[m for m in transform if m.exportable()])


def init_window(wnd: str, frame_len: int, device: th.device = "cpu") -> th.Tensor:
    """window coefficients; periodic (librosa convention) for everything but "rect" """
    if wnd not in _WINDOWS:
        raise RuntimeError(f"Unknown window type: {wnd}")
    if wnd == "rect":
        c = th.ones(frame_len)
    elif wnd == "sqrthann":
        c = th.hann_window(frame_len, periodic=True)**0.5
    else:
        fn = {"hann": th.hann_window, "hamm": th.hamming_window, "blackman": th.blackman_window,
              "bartlett": th.bartlett_window}[wnd]
        c = fn(frame_len, periodic=True)
    return c.to(device)


def _fft_size(frame_len: int, round_pow_of_two: bool, mode: str) -> int:
    if round_pow_of_two or mode == "kaldi":
        return 2**math.ceil(math.log2(frame_len))
    return frame_len


def init_kernel(frame_len: int,
                frame_hop: int,
                window: th.Tensor,
                round_pow_of_two: bool = True,
                normalized: bool = False,
                inverse: bool = False,
                mode: str = "librosa") -> Tuple[th.Tensor, th.Tensor]:
    """(K [2W,1,L], w [L]) exactly as the reference stores them in `state_dict`s.  The HIP path
    never reads K; it exists for checkpoint compatibility (strict load_state_dict)."""
    if mode not in ["librosa", "kaldi"]:
        raise ValueError(f"Unsupported mode: {mode}")
    W = _fft_size(frame_len, round_pow_of_two, mode)
    if mode == "librosa" and W != frame_len:
        lpad = (W - frame_len) // 2
        window = tf.pad(window, (lpad, W - frame_len - lpad))
    S = W**0.5 if normalized else 1
    spec = th.fft.fft(th.eye(W) / S, dim=-1)
    K = th.stack([spec.real, spec.imag], dim=-1)
    if mode == "kaldi":
        K = K[:frame_len]
    if inverse and not normalized:
        K = K / W
    K = K.transpose(0, 2).reshape(W * 2, 1, -1)
    return K.to(window.device), window


def _htk_mel_matrix(sr, n_fft, n_mels, fmin, fmax, slaney_norm) -> np.ndarray:
    """librosa.filters.mel(..., htk=True) from its published definition (librosa is optional here)"""
    to_mel = lambda hz: 2595.0 * np.log10(1.0 + np.asarray(hz, dtype=np.float64) / 700.0)
    to_hz = lambda mel: 700.0 * (10.0**(np.asarray(mel, dtype=np.float64) / 2595.0) - 1.0)
    bins = np.linspace(0, float(sr) / 2, 1 + n_fft // 2)
    edges = to_hz(np.linspace(to_mel(fmin), to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    dist = edges[:, None] - bins[None, :]
    rise = -dist[:-2] / width[:-1, None]
    fall = dist[2:] / width[1:, None]
    weights = np.maximum(0, np.minimum(rise, fall)).astype(np.float32)
    if slaney_norm:
        # librosa scales its float32 matrix in place by the float64 band norms
        weights *= (2.0 / (edges[2:n_mels + 2] - edges[:n_mels]))[:, None]
    return weights


def speed_perturb_filter(src_sr: int, dst_sr: int, cutoff_ratio: float = 0.95,
                         num_zeros: int = 64) -> th.Tensor:
    """Polyphase windowed-sinc resampling filter bank [dst, src, 2 pad + 1] of the speed
    perturbation layer (utils.py:159-190, after lilfilter's resampler): tap (d, s, k) sits at time
    d / dst - s / src - k + pad (in source blocks), Hann window over +- pad, low-pass at
    cutoff_ratio x the lower rate.  Frozen parameters of SpeedPerturbTransform."""
    if src_sr == dst_sr:
        raise ValueError(f"src_sr should not be equal to dst_sr: {src_sr}/{dst_sr}")
    gcd = math.gcd(src_sr, dst_sr)
    src, dst = src_sr // gcd, dst_sr // gcd
    if src == 1 or dst == 1:
        raise ValueError("do not support integer downsample/upsample")
    zeros_per_block = min(src, dst) * cutoff_ratio
    pad = 1 + int(num_zeros / zeros_per_block)
    t = (np.arange(dst)[:, None, None] / float(dst) - np.arange(src)[None, :, None] / float(src) -
         np.arange(2 * pad + 1)[None, None, :] + pad)
    hann = np.heaviside(1 - np.abs(t / pad), 0.0) * (0.5 + 0.5 * np.cos(t / pad * math.pi))
    return th.tensor(np.sinc(t * zeros_per_block) * hann * zeros_per_block / float(src),
                     dtype=th.float32)


def mel_filter(frame_len: int,
               round_pow_of_two: bool = True,
               num_bins: Optional[int] = None,
               sr: int = 16000,
               num_mels: int = 80,
               fmin: float = 0.0,
               fmax: Optional[float] = None,
               norm: bool = False) -> th.Tensor:
    """mel filter coefficients, num_mels x (N/2+1); librosa when installed, else its algorithm"""
    if num_bins is None:
        N = 2**math.ceil(math.log2(frame_len)) if round_pow_of_two else frame_len
    else:
        N = (num_bins - 1) * 2
    freq_upper = sr // 2
    if fmax is None:
        fmax = freq_upper
    else:
        fmax = min(fmax + freq_upper if fmax < 0 else fmax, freq_upper)
    fmin = max(0, fmin)
    try:
        import librosa.filters as filters
        mel = filters.mel(sr=sr, n_fft=N, n_mels=num_mels, fmax=fmax, fmin=fmin, htk=True,
                          norm="slaney" if norm else None)
    except ImportError:
        mel = _htk_mel_matrix(sr, N, num_mels, fmin, fmax, norm)
    return th.tensor(mel, dtype=th.float32)


# --------------------------------------------------------------------------------------------
# kernel launchers
# --------------------------------------------------------------------------------------------
def _stft_params(fft_size, frame_len, frame_hop, onesided, center, polar, pre_emphasis, eps,
                 scale) -> nat.StftParams:
    return nat.StftParams(fft_size, frame_len, frame_hop, fft_size // 2 + 1 if onesided else
                          fft_size, int(center), int(polar), float(pre_emphasis), float(eps),
                          float(scale))


class _StftFunction(th.autograd.Function):
    """STFT with its adjoint as backward (aps_stft_backward); window / DFT basis are constants"""

    @staticmethod
    def forward(ctx, wav, window, fft_size, frame_hop, onesided, center, normalized):
        ctx.save_for_backward(window)
        ctx.cfg = (fft_size, frame_hop, onesided, center, normalized, tuple(wav.shape))
        return stft_to_store(wav, window, fft_size, frame_hop, onesided=onesided, center=center,
                             normalized=normalized)

    @staticmethod
    def backward(ctx, grad_store):
        (window,) = ctx.saved_tensors
        fft_size, frame_hop, onesided, center, normalized, shape = ctx.cfg
        lib = nat.load()
        g = nat.f32c(grad_store)
        S, L = shape[-1], window.shape[0]
        T, F = g.shape[-3], g.shape[-2]
        scale = 1.0 / math.sqrt(fft_size) if normalized else 1.0
        p = _stft_params(fft_size, L, frame_hop, onesided, center, False, 0, EPSILON, scale)
        num_seq = g.numel() // (T * F * 2)
        grad_wav = th.empty(shape, device=g.device, dtype=th.float32)
        work = th.empty(num_seq * T * L, device=g.device, dtype=th.float32)
        rc = lib.aps_stft_backward(nat.ptr(g), num_seq, T, T * F * 2, F * 2,
                                   nat.ptr(nat.f32c(window)), C.byref(p), nat.ptr(grad_wav), S,
                                   nat.ptr(work), nat.stream_of(g))
        nat.check(rc, "aps_stft_backward")
        return grad_wav, None, None, None, None, None, None


class _IstftFunction(th.autograd.Function):
    """iSTFT with its adjoint as backward (aps_stft_inverse_backward)"""

    @staticmethod
    def forward(ctx, store, window, fft_size, frame_hop, onesided, center, normalized, eps):
        ctx.save_for_backward(window)
        ctx.cfg = (fft_size, frame_hop, onesided, center, normalized, eps, tuple(store.shape))
        return istft_from_store(store, window, fft_size, frame_hop, onesided=onesided,
                                center=center, normalized=normalized, eps=eps)

    @staticmethod
    def backward(ctx, grad_wav):
        (window,) = ctx.saved_tensors
        fft_size, frame_hop, onesided, center, normalized, eps, shape = ctx.cfg
        lib = nat.load()
        g = nat.f32c(grad_wav)
        N, T, F, _ = shape
        L = window.shape[0]
        scale = 1.0 / math.sqrt(fft_size) if normalized else 1.0 / fft_size
        p = _stft_params(fft_size, L, frame_hop, onesided, center, False, 0, eps, scale)
        grad_store = th.empty(shape, device=g.device, dtype=th.float32)
        work = th.empty(N * ((T - 1) * frame_hop + L), device=g.device, dtype=th.float32)
        rc = lib.aps_stft_inverse_backward(nat.ptr(g), N, g.shape[-1], nat.ptr(nat.f32c(window)),
                                           C.byref(p), nat.ptr(grad_store), T * F * 2, F * 2, T,
                                           nat.ptr(work), nat.stream_of(g))
        nat.check(rc, "aps_stft_inverse_backward")
        return grad_store, None, None, None, None, None, None, None


def stft_to_store(wav: th.Tensor, window: th.Tensor, fft_size: int, frame_hop: int,
                  onesided: bool = True, center: bool = False, polar: bool = False,
                  pre_emphasis: float = 0, normalized: bool = False,
                  eps: float = EPSILON) -> th.Tensor:
    """wav N x (C) x S -> store N x (C) x T x F x 2 (one launch, K1 of SURVEY 2.4)"""
    if wav.dim() not in [2, 3]:
        raise RuntimeError(f"STFT expect 2D/3D tensor, but got {wav.dim():d}D")
    if nat.needs_grad(wav):
        if polar or pre_emphasis > 0:
            raise NotImplementedError("aps_amd: backward through the polar / pre-emphasised STFT "
                                      "is not implemented (rectangular STFT only)")
        return _StftFunction.apply(wav, window.detach(), fft_size, frame_hop, onesided, center,
                                   normalized)
    nat.require_device(wav, window)
    lib = nat.load()
    pcm16 = wav.dtype == th.int16   # int16 PCM: samples are value / 32768, formed in the kernel (aps/io/audio.py:41-44)
    wav = _pcm16c(wav) if pcm16 else nat.f32c(wav)
    S = wav.shape[-1]
    L = window.shape[0]
    scale = 1.0 / math.sqrt(fft_size) if normalized else 1.0
    p = _stft_params(fft_size, L, frame_hop, onesided, center, polar, pre_emphasis, eps, scale)
    T = int(lib.aps_stft_num_frames(S, C.byref(p)))
    if T <= 0:
        raise RuntimeError(f"signal of {S} samples is shorter than one frame ({L})")
    store = alloc_store(tuple(wav.shape[:-1]), T, p.num_bins, wav.device)
    num_seq = wav.numel() // S
    entry = lib.aps_stft_forward_pcm16 if pcm16 else lib.aps_stft_forward
    rc = entry(nat.ptr(wav), num_seq, S, nat.ptr(nat.f32c(window)), C.byref(p),
               nat.ptr(store), T * p.num_bins * 2, p.num_bins * 2, T, nat.stream_of(wav))
    nat.check(rc, "aps_stft_forward_pcm16" if pcm16 else "aps_stft_forward")
    return store


def _pcm16c(wav: th.Tensor) -> th.Tensor:
    """int16 PCM as the *_pcm16 entry points take it: contiguous, 4-byte aligned"""
    wav = wav.detach().contiguous()
    if wav.data_ptr() % 4:
        wav = wav.clone()
    return wav


def stft_features(wav: th.Tensor, window: th.Tensor, fft_size: int, frame_hop: int, plan,
                  ref_channel: int = 0, pairs=None, ipd_sin: bool = False, center: bool = False,
                  pre_emphasis: float = 0, normalized: bool = False, write_store: bool = True,
                  nan_flag: Optional[th.Tensor] = None):
    """wav N x C x S -> (store N x C x T x F x 2 | None, feats N x T x D) in ONE launch, or None
    when the configuration is outside the fused kernel's domain (caller then runs two kernels)."""
    from aps_amd.ops import _feat_params, _mel_ptrs, _pair_tensors
    if wav.dim() != 3 or fft_size != 512 or frame_hop % 2 or window.shape[0] % 2:
        return None
    N, Cn, S = wav.shape
    if Cn > 8 or (plan is not None and plan.mel is not None and pairs is not None):
        return None
    if pre_emphasis > 0 and (Cn != 1 or write_store):
        return None
    if not write_store and Cn != 1:
        return None
    nat.require_device(wav, window)
    lib = nat.load()
    pcm16 = wav.dtype == th.int16
    wav = _pcm16c(wav) if pcm16 else nat.f32c(wav)
    L = window.shape[0]
    scale = 1.0 / math.sqrt(fft_size) if normalized else 1.0
    p = _stft_params(fft_size, L, frame_hop, True, center, False, pre_emphasis, EPSILON, scale)
    T = int(lib.aps_stft_num_frames(S, C.byref(p)))
    if T <= 0:
        raise RuntimeError(f"signal of {S} samples is shorter than one frame ({L})")
    F = p.num_bins
    num_pairs, pl, pr = 0, None, None
    if pairs is not None:
        il, ir = pairs
        if Cn < 2 or max(il + ir) >= Cn or min(il + ir) < 0:
            raise RuntimeError(f"IPD pair index out of range for {Cn} channels: {il} / {ir}")
        num_pairs = len(il)
        pl, pr = _pair_tensors(tuple(il), tuple(ir), wav.device)
    ref = ref_channel if plan is not None else -1
    q = _feat_params(F, Cn, ref, plan, num_pairs, ipd_sin)
    D0 = 0 if plan is None else (plan.mel.num_mels if plan.mel else F)
    D = D0 + num_pairs * (2 if ipd_sin else 1) * F
    store = alloc_store((N, Cn), T, F, wav.device) if write_store else None
    feats = th.empty(N, T, D, device=wav.device, dtype=th.float32)
    ms, ml, mo, mw = _mel_ptrs(plan)
    entry = lib.aps_stft_features_pcm16 if pcm16 else lib.aps_stft_features
    rc = entry(nat.ptr(wav), N, Cn, S, nat.ptr(nat.f32c(window)), C.byref(p),
               C.byref(q), ms, ml, mo, mw, nat.ptr(pl), nat.ptr(pr),
               nat.ptr(store), T * F * 2, F * 2, T, nat.ptr(feats),
               nat.ptr(nan_flag), nat.stream_of(wav))
    if rc == -2:  # APS_ERR_UNSUPPORTED: not the fused kernel's domain
        return None
    nat.check(rc, "aps_stft_features")
    return store, feats


def istft_from_store(store: th.Tensor, window: th.Tensor, fft_size: int, frame_hop: int,
                     onesided: bool = True, center: bool = False, polar: bool = False,
                     normalized: bool = False, eps: float = EPSILON) -> th.Tensor:
    """store N x T x F x 2 -> wav N x S (K12 of SURVEY 2.4)"""
    if nat.needs_grad(store):
        if polar:
            raise NotImplementedError("aps_amd: backward through the polar iSTFT is not implemented")
        return _IstftFunction.apply(store, window.detach(), fft_size, frame_hop, onesided, center,
                                    normalized, eps)
    nat.require_device(store, window)
    lib = nat.load()
    N, T, F, _ = store.shape
    L = window.shape[0]
    scale = 1.0 / math.sqrt(fft_size) if normalized else 1.0 / fft_size
    p = _stft_params(fft_size, L, frame_hop, onesided, center, polar, 0, eps, scale)
    if F != p.num_bins:
        raise RuntimeError(f"iSTFT expects {p.num_bins} bins, got {F}")
    crop = L // 2 if center else 0
    S = (T - 1) * frame_hop + L - 2 * crop
    wav = th.empty(N, S, device=store.device, dtype=th.float32)
    work = th.empty(N * T * L, device=store.device, dtype=th.float32)
    rc = lib.aps_stft_inverse(nat.ptr(store), N, T, store.stride(0), store.stride(1),
                              nat.ptr(nat.f32c(window)), C.byref(p), nat.ptr(wav), S,
                              nat.ptr(work), nat.stream_of(store))
    nat.check(rc, "aps_stft_inverse")
    return wav


def _window_for(window: str, frame_len: int, round_pow_of_two: bool, mode: str, device):
    """window as the kernels want it: the module's `w` (centre padded to W in librosa mode)"""
    w = init_window(window, frame_len, device=device)
    kmode = "librosa" if mode == "torch" else mode
    W = _fft_size(frame_len, round_pow_of_two, kmode)
    if kmode == "librosa" and W != frame_len:
        lpad = (W - frame_len) // 2
        w = tf.pad(w, (lpad, W - frame_len - lpad))
    return w, W


def forward_stft(wav: th.Tensor,
                 frame_len: int,
                 frame_hop: int,
                 window: str = "sqrthann",
                 round_pow_of_two: bool = True,
                 return_polar: bool = False,
                 pre_emphasis: float = 0,
                 normalized: bool = False,
                 onesided: bool = True,
                 center: bool = False,
                 mode: str = "librosa",
                 eps: float = EPSILON) -> th.Tensor:
    """functional STFT, N x (C) x S -> N x (C) x F x T x 2"""
    if mode not in ["librosa", "kaldi", "torch"]:
        raise ValueError(f"Unsupported mode: {mode}")
    w, W = _window_for(window, frame_len, round_pow_of_two, mode, wav.device)
    store = stft_to_store(wav, w, W, frame_hop, onesided=onesided, center=center,
                          polar=return_polar, pre_emphasis=0 if mode == "torch" else pre_emphasis,
                          normalized=normalized, eps=eps)
    return packed_view(store)


def _as_4d_store(transform: th.Tensor) -> th.Tensor:
    if transform.dim() == 3:
        transform = transform[None]
    if transform.dim() != 4:
        raise RuntimeError(f"Expect 4D tensor, but got {transform.dim()}D")
    return store_of(transform)


def inverse_stft(transform: th.Tensor,
                 frame_len: int,
                 frame_hop: int,
                 return_polar: bool = False,
                 window: str = "sqrthann",
                 round_pow_of_two: bool = True,
                 normalized: bool = False,
                 onesided: bool = True,
                 center: bool = False,
                 mode: str = "librosa",
                 eps: float = EPSILON) -> th.Tensor:
    """functional iSTFT, (N) x F x T x 2 -> N x S"""
    if mode not in ["librosa", "kaldi", "torch"]:
        raise ValueError(f"Unsupported mode: {mode}")
    w, W = _window_for(window, frame_len, round_pow_of_two, mode, transform.device)
    return istft_from_store(_as_4d_store(transform), w, W, frame_hop, onesided=onesided,
                            center=center, polar=return_polar, normalized=normalized, eps=eps)


# --------------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------------
class STFTBase(nn.Module):
    """Base layer for (i)STFT: same ctor, attributes and frozen parameters (`K`, `w`) as the
    reference (utils.py:594-675)."""

    def __init__(self,
                 frame_len: int,
                 frame_hop: int,
                 window: str = "sqrthann",
                 round_pow_of_two: bool = True,
                 normalized: bool = False,
                 pre_emphasis: float = 0,
                 onesided: bool = True,
                 inverse: bool = False,
                 center: bool = False,
                 mode: str = "librosa") -> None:
        super(STFTBase, self).__init__()
        if mode not in ["librosa", "kaldi", "torch"]:
            raise ValueError(f"Unsupported mode: {mode}")
        if mode != "torch":
            K, w = init_kernel(frame_len, frame_hop, init_window(window, frame_len),
                               round_pow_of_two=round_pow_of_two, normalized=normalized,
                               inverse=inverse, mode=mode)
            self.K = nn.Parameter(K, requires_grad=False)
            self.w = nn.Parameter(w, requires_grad=False)
            self.num_bins = self.K.shape[0] // 4 + 1
            self.pre_emphasis = pre_emphasis
            self.win_length = self.K.shape[2]
            self.fft_size = self.K.shape[0] // 2
        else:
            self.K = None
            self.w = nn.Parameter(init_window(window, frame_len), requires_grad=False)
            self.fft_size = _fft_size(frame_len, round_pow_of_two, "librosa")
            self.num_bins = self.fft_size // 2 + 1
            self.pre_emphasis = 0
            self.win_length = self.fft_size
        self.frame_len = frame_len
        self.frame_hop = frame_hop
        self.window = window
        self.normalized = normalized
        self.onesided = onesided
        self.center = center
        self.mode = mode

    def _kernel_window(self) -> th.Tensor:
        """[L] window for the kernels (torch mode keeps the un-padded window as its parameter)"""
        w = self.w.data
        if self.mode == "torch" and w.shape[0] != self.fft_size:
            lpad = (self.fft_size - w.shape[0]) // 2
            w = tf.pad(w, (lpad, self.fft_size - w.shape[0] - lpad))
        return w

    def num_frames(self, wav_len: th.Tensor) -> th.Tensor:
        """number of frames per utterance; integer exact (utils.py:653-662)"""
        # the reference's sanity check reads the result on the host (a sync); it cannot run while
        # the stream is being captured into a graph
        if not (wav_len.is_cuda and th.cuda.is_current_stream_capturing()):
            assert th.sum(wav_len <= self.win_length) == 0
        from aps_amd.ops import length_map
        return length_map(wav_len, 0 if self.center else -self.win_length, self.frame_hop, 1)

    def extra_repr(self) -> str:
        str_repr = (f"num_bins={self.num_bins}, win_length={self.win_length}, " +
                    f"stride={self.frame_hop}, window={self.window}, " +
                    f"center={self.center}, mode={self.mode}")
        if not self.onesided:
            str_repr += f", onesided={self.onesided}"
        if self.pre_emphasis > 0:
            str_repr += f", pre_emphasis={self.pre_emphasis}"
        if self.normalized:
            str_repr += f", normalized={self.normalized}"
        return str_repr


class STFT(STFTBase):
    """Short-time Fourier Transform as a layer (utils.py:678-717)"""

    def __init__(self, *args, **kwargs):
        super(STFT, self).__init__(*args, inverse=False, **kwargs)

    def to_store(self, wav: th.Tensor, return_polar: bool = False,
                 eps: float = EPSILON) -> th.Tensor:
        """N x (C) x S -> bin-fastest store N x (C) x T x F x 2"""
        return stft_to_store(wav, self._kernel_window(), self.fft_size, self.frame_hop,
                             onesided=self.onesided, center=self.center, polar=return_polar,
                             pre_emphasis=self.pre_emphasis, normalized=self.normalized, eps=eps)

    def forward(self, wav: th.Tensor, return_polar: bool = False,
                eps: float = EPSILON) -> th.Tensor:
        """N x (C) x S -> N x (C) x F x T x 2 (view of the store)"""
        return packed_view(self.to_store(wav, return_polar=return_polar, eps=eps))


class iSTFT(STFTBase):
    """Inverse Short-time Fourier Transform as a layer (utils.py:720-758)"""

    def __init__(self, *args, **kwargs):
        super(iSTFT, self).__init__(*args, inverse=True, **kwargs)

    def forward(self, transform: th.Tensor, return_polar: bool = False,
                eps: float = EPSILON) -> th.Tensor:
        """(N) x F x T x 2 -> N x S"""
        return istft_from_store(_as_4d_store(transform), self._kernel_window(), self.fft_size,
                                self.frame_hop, onesided=self.onesided, center=self.center,
                                polar=return_polar, normalized=self.normalized, eps=eps)
