"""
Spectrogram store: the HBM layout every kernel of this build reads and writes.

    store[n, (c,) t, f, 0:2]      fp32, bin axis fastest, (re, im) interleaved

A frame's FFT produces all bins of one t, features / masks / beamformer outputs are N x T x F, and
covariance reduces over t per bin -- so with the bin axis fastest every kernel's global access is
a contiguous run of bins and no pass transposes.  The reference layout N x (C) x F x T x 2
(aps/transform/utils.py:290, produced by conv1d) is exposed as the transposed VIEW of the store:
identical shape and values, different strides.
"""
from typing import Tuple

import torch as th


def alloc_store(lead: Tuple[int, ...], T: int, F: int, device) -> th.Tensor:
    return th.empty(*lead, T, F, 2, device=device, dtype=th.float32)


def packed_view(store: th.Tensor) -> th.Tensor:
    """store (..., T, F, 2) -> reference-shaped view (..., F, T, 2)"""
    return store.transpose(-2, -3)


def _usable(v: th.Tensor) -> bool:
    return v.dtype == th.float32 and v.stride(-1) == 1 and v.stride(-2) == 2


def store_of(packed: th.Tensor) -> th.Tensor:
    """reference-shaped (..., F, T, 2) tensor -> (..., T, F, 2) tensor the kernels can address
    (a view when `packed` came from this build, one transposing copy for foreign tensors)."""
    v = packed.transpose(-2, -3)
    if _usable(v):
        return v
    return v.float().contiguous()


def store_of_pair(real: th.Tensor, imag: th.Tensor) -> th.Tensor:
    """ComplexTensor halves (..., F, T) -> (..., T, F, 2) store; zero-copy when the halves are the
    [..., 0] / [..., 1] views of one of our stores (the EnhASRBase call pattern, enh_att.py:85)."""
    if (real.dtype == th.float32 and imag.dtype == th.float32 and real.shape == imag.shape and
            real.stride() == imag.stride() and real.stride(-2) == 2 and
            real.untyped_storage().data_ptr() == imag.untyped_storage().data_ptr() and
            imag.storage_offset() == real.storage_offset() + 1):
        rt = real.transpose(-1, -2)
        return th.as_strided(real, (*rt.shape, 2), (*rt.stride(), 1), real.storage_offset())
    return th.stack([real, imag], -1).transpose(-2, -3).float().contiguous()
