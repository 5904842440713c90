"""
Frame-by-frame (i)STFT -- the surface of aps/transform/streaming.py (StreamingSTFT /
StreamingiSTFT: `step` for one frame, `forward` for a whole signal, `reset` / `flush` for the
overlap caches).  The transform of a frame is the STFT / iSTFT kernel launched for one frame
(csrc/stft.hip: any FFT size <= 4096, i.e. also the 400-point frames of the "kaldi" mode);
`forward` is ONE launch over all frames instead of the reference's Python loop.  The overlap caches
of StreamingiSTFT.step are win_length-long vectors updated with a handful of elementwise torch
ops: a stateful, latency-bound API by construction.
"""
import torch as th

from aps_amd.const import EPSILON
from aps_amd.spectrogram import packed_view
from aps_amd.transform.utils import STFTBase, _as_4d_store, istft_from_store, stft_to_store


class _StreamingBase(STFTBase):

    def _frame_window(self) -> th.Tensor:
        # the reference multiplies a win_length frame by `w`: both must have that length
        # (streaming.py:33, 98), which rules out mode="torch" with a rounded-up FFT size
        w = self.w.data
        if w.shape[0] != self.win_length:
            raise RuntimeError(f"window of {w.shape[0]} samples does not match the frame length "
                               f"{self.win_length} (mode={self.mode})")
        return w


class StreamingSTFT(_StreamingBase):
    """To mimic streaming (frame by frame processing) STFT (streaming.py:13-64)"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, inverse=False, **kwargs)

    def _store(self, wav: th.Tensor, return_polar: bool, eps: float) -> th.Tensor:
        return stft_to_store(wav, self._frame_window(), self.win_length, self.frame_hop,
                             onesided=True, center=False, polar=return_polar, pre_emphasis=0,
                             normalized=self.normalized, eps=eps)

    def step(self, frame: th.Tensor, return_polar: bool = False,
             eps: float = EPSILON) -> th.Tensor:
        """one frame: N x (C) x S (S = win_length) -> N x (C) x F x 2"""
        if frame.shape[-1] != self.win_length:
            raise RuntimeError(f"a frame has {self.win_length} samples, got {frame.shape[-1]}")
        return self._store(frame, return_polar, eps).squeeze(-3)

    def forward(self, wav: th.Tensor, return_polar: bool = False,
                eps: float = EPSILON) -> th.Tensor:
        """N x (C) x S -> N x (C) x F x T x 2, frames at 0, hop, 2 hop, ... (no padding)"""
        return packed_view(self._store(wav, return_polar, eps))


class StreamingiSTFT(_StreamingBase):
    """To mimic streaming (frame by frame processing) iSTFT (streaming.py:67-152)"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, inverse=True, **kwargs)
        self.reset()

    def reset(self):
        overlap = self.win_length - self.frame_hop
        self.wav_cache = th.zeros(overlap, device=self.w.device)
        self.win_cache = th.zeros(overlap, device=self.w.device)

    def _windowed_frame(self, frame: th.Tensor, return_polar: bool) -> th.Tensor:
        """N x F x 2 -> irfft(frame) * w, N x W: the inverse kernel run for a single frame leaves
        exactly that in its workspace (include/aps_amd.h: "windowed frames before OLA")"""
        import ctypes as C
        import math
        from aps_amd import _native as nat
        from aps_amd.transform.utils import _stft_params
        w = self._frame_window()
        nat.require_device(frame, w)
        lib = nat.load()
        store = nat.f32c(frame).unsqueeze(1)  # N x 1 x F x 2
        N, _, F, _ = store.shape
        W = self.win_length
        scale = 1.0 / math.sqrt(W) if self.normalized else 1.0 / W
        p = _stft_params(W, W, self.frame_hop, True, False, return_polar, 0, EPSILON, scale)
        if F != p.num_bins:
            raise RuntimeError(f"iSTFT expects {p.num_bins} bins, got {F}")
        wav = th.empty(N, W, device=store.device, dtype=th.float32)
        work = th.empty(N, W, device=store.device, dtype=th.float32)
        rc = lib.aps_stft_inverse(nat.ptr(store), N, 1, store.stride(0), store.stride(1),
                                  nat.ptr(nat.f32c(w)), C.byref(p), nat.ptr(wav), W, nat.ptr(work),
                                  nat.stream_of(store))
        nat.check(rc, "aps_stft_inverse")
        return work

    def step(self, frame: th.Tensor, return_polar: bool = False,
             eps: float = EPSILON) -> th.Tensor:
        """one frame: N x F x 2 -> the next frame_hop finished samples, N x hop"""
        return self.norm(self._windowed_frame(frame, return_polar), eps=eps)

    def norm(self, frame: th.Tensor, eps: float = EPSILON) -> th.Tensor:
        """overlap-add with the cached tail and normalise by the accumulated window energy"""
        if self.wav_cache.device != frame.device:
            self.wav_cache = self.wav_cache.to(frame.device)
            self.win_cache = self.win_cache.to(frame.device)
        window = (self._frame_window() ** 2).clone()
        overlap = self.win_cache.shape[0]
        frame[:, :overlap] += self.wav_cache
        window[:overlap] += self.win_cache
        self.win_cache = window[self.frame_hop:]
        self.wav_cache = frame[:, self.frame_hop:]
        frame = frame / (window + eps)
        return frame[:, :self.frame_hop]

    def flush(self, eps: float = EPSILON) -> th.Tensor:
        return self.wav_cache / (self.win_cache + eps)

    def forward(self, transform: th.Tensor, return_polar: bool = False,
                eps: float = EPSILON) -> th.Tensor:
        """N x F x T x 2 -> N x S with S = T hop + (win_length - hop): all steps and the flush in
        one launch (overlap-add of every frame, window-energy normaliser + eps, no crop)"""
        self.reset()
        return istft_from_store(_as_4d_store(transform), self._frame_window(), self.win_length,
                                self.frame_hop, onesided=True, center=False, polar=return_polar,
                                normalized=self.normalized, eps=eps)
