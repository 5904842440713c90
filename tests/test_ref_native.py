"""
CPU: the oracle against the REFERENCE's own native code (csrc/utils/{fft,window,stft}.cc compiled
in place into oracle/_ref/libaps_ref.so by oracle/Makefile).  Independent pin for window values,
pow-2 rounding, librosa/kaldi window placement, framing and the packed RealFFT layout.
Skipped when the library has not been built (it needs /root/reference at build time).
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import aps_oracle as orc
from tests.conftest import ROOT, golden, assert_close

LIB = os.path.join(ROOT, "oracle", "_ref", "libaps_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def ref():
    lib = C.CDLL(LIB)
    lib.aps_ref_stft.restype = C.c_int32
    lib.aps_ref_fft_size.restype = C.c_int32
    lib.aps_ref_frame_length.restype = C.c_int32
    return lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


@pytest.mark.parametrize("name", ["hann", "sqrthann", "hamm", "rect", "blackman", "bartlett"])
@pytest.mark.parametrize("n", [256, 400, 512])
def test_windows_match_native(ref, name, n):
    out = np.zeros(n, dtype=np.float32)
    ref.aps_ref_window(name.encode(), n, 1, _fp(out))
    # tests/csrc/test-utils-stft.cc:12-31 uses 1e-4 against torch; we are much closer
    np.testing.assert_allclose(orc.window(name, n).numpy(), out, rtol=0, atol=2e-6)


def test_pow2_rounding_and_frame_length(ref):
    for fl, mode in [(400, "librosa"), (400, "kaldi"), (512, "librosa"), (256, "kaldi")]:
        W = ref.aps_ref_fft_size(fl, 160, b"hann", mode.encode())
        L = ref.aps_ref_frame_length(fl, 160, b"hann", mode.encode())
        assert W == orc.fft_size_of(fl, True, mode)
        assert L == (W if mode == "librosa" else fl)


def test_complex_fft_matches_dft(ref):
    """the reference's radix-2 ComplexFFT (the part its own tests/csrc/test-fft.cc exercises)"""
    rng = np.random.default_rng(0)
    for n in [64, 256, 512]:
        z = rng.standard_normal(2 * n).astype(np.float32)
        buf = z.copy()
        ref.aps_ref_complex_fft(_fp(buf), 2 * n, 0)
        want = np.fft.fft(z[0::2].astype(np.float64) + 1j * z[1::2])
        got = buf[0::2] + 1j * buf[1::2]
        assert np.abs(got - want).max() / np.abs(want).max() < 2e-6
        ref.aps_ref_complex_fft(_fp(buf), 2 * n, 1)
        np.testing.assert_allclose(buf, z, atol=2e-6)


def test_native_realfft_is_not_a_spectrum_oracle(ref):
    """Finding: FFTComputer::RealFFT(invert=false) does not return the DFT of its
    input for most bins (the reference's own TestRealFFT is not run by its test main,
    tests/csrc/test-fft.cc:76-80).  Only DC / Nyquist and the forward->inverse round trip are
    usable, so spectra are pinned by the golden vectors of the PyTorch path instead."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal(512).astype(np.float32)
    buf = x.copy()
    ref.aps_ref_real_fft(_fp(buf), 512, 0)
    X = np.fft.rfft(x.astype(np.float64))
    assert abs(buf[0] - X[0].real) < 1e-3 and abs(buf[1] - X[256].real) < 1e-3
    ref.aps_ref_real_fft(_fp(buf), 512, 1)
    np.testing.assert_allclose(buf, x, atol=1e-5)  # round trip holds


def test_framing_matches_native(ref):
    wav = golden("stft_egs1_512_sqrthann")["wav"][0].numpy().copy()
    for fl, fh, mode in [(512, 256, "librosa"), (400, 160, "librosa"), (400, 160, "kaldi")]:
        W = orc.fft_size_of(fl, True, mode)
        L = W if mode == "librosa" else fl
        T = orc.num_frames(len(wav), L, fh, False)
        out = np.zeros((T + 4, W), dtype=np.float32)
        n = ref.aps_ref_stft(_fp(wav), len(wav), fl, fh, b"hann", mode.encode(), _fp(out), T + 4)
        assert n == T  # frame count: exact


def test_istft_roundtrip_native(ref):
    wav = golden("stft_egs1_512_sqrthann")["wav"][0].numpy().copy()
    fl, fh = 512, 256
    T = orc.num_frames(len(wav), fl, fh, False)
    spec = np.zeros((T, fl), dtype=np.float32)
    ref.aps_ref_stft(_fp(wav), len(wav), fl, fh, b"sqrthann", b"librosa", _fp(spec), T)
    out = np.zeros((T - 1) * fh + fl, dtype=np.float32)
    ref.aps_ref_istft(_fp(spec), T, fl, fh, b"sqrthann", b"librosa", _fp(out))
    mine = orc.istft(orc.stft(torch.from_numpy(wav)[None], fl, fh), fl, fh)[0]
    # both reconstruct the signal wherever the window^2 overlap-add normaliser is well conditioned
    n = len(out)
    assert_close(torch.from_numpy(out[fh:n - fh]), torch.from_numpy(wav[fh:n - fh]), 1e-4)
    assert_close(mine[fh:n - fh], torch.from_numpy(wav[fh:n - fh]), 1e-4)
