"""
GPU: ComplexTensor.__matmul__ / __rmatmul__ / inverse on the HIP kernels (aps_cplx_matmul,
aps_cplx_inverse), mirroring the reference's own self-tests (aps/cplx.py:301-364: random operands,
matmul with complex and real operands on either side, inverse) against numpy complex arithmetic, plus
the shapes the MVDR chain produces (N x F x C x C covariances).
"""
import numpy as np
import pytest
import torch

from aps_amd.cplx import ComplexTensor

pytestmark = pytest.mark.gpu


def _rand(rng, shape, cplx=True):
    re = rng.random(shape).astype(np.float32)
    if not cplx:
        return torch.from_numpy(re), re
    im = rng.random(shape).astype(np.float32)
    return ComplexTensor(torch.from_numpy(re), torch.from_numpy(im)), re + 1j * im


def _same(got: ComplexTensor, want, tol=2e-5):
    g = got.real.cpu().double().numpy() + 1j * got.imag.cpu().double().numpy()
    assert g.shape == want.shape, (g.shape, want.shape)
    assert np.abs(g - want).max() <= tol * max(np.abs(want).max(), 1e-30)


def _dev(x, device):
    return ComplexTensor(x.real.to(device), x.imag.to(device)) if isinstance(x, ComplexTensor) else x.to(device)


@pytest.mark.parametrize("shape_a,shape_b", [((5, 5), (5, 5)), ((3, 4, 5), (3, 5, 2)), ((2, 7, 4, 4), (2, 7, 4, 4)),
                                             ((6, 4, 8), (8, 3)), ((32, 257, 4, 4), (32, 257, 4, 1))])
def test_matmul_runs_on_the_hip_kernel(device, shape_a, shape_b, monkeypatch):
    rng = np.random.default_rng(len(shape_a) + shape_a[-1])
    a, na = _rand(rng, shape_a)
    b, nb = _rand(rng, shape_b)
    r, nr = _rand(rng, shape_b, cplx=False)
    calls = []
    import aps_amd.cplx as cplx_mod
    real = cplx_mod._hip_matmul
    monkeypatch.setattr(cplx_mod, "_hip_matmul", lambda *args: calls.append(1) or real(*args))
    ad, bd, rd = _dev(a, device), _dev(b, device), _dev(r, device)
    _same(ad @ bd, na @ nb)                 # complex @ complex (cplx.py:242-252)
    _same(ad @ rd, na @ nr)                 # complex @ real
    l, nl = _rand(rng, shape_a, cplx=False)
    _same(_dev(l, device) @ bd, nl @ nb)    # real @ complex (__rmatmul__, cplx.py:255-266)
    assert len(calls) == 3
    out = ad @ bd
    assert out.real.is_cuda and out.real.dtype == torch.float32


@pytest.mark.parametrize("C", [1, 2, 3, 4, 6, 8])
def test_inverse_runs_on_the_hip_kernel(device, C):
    rng = np.random.default_rng(C)
    shape = (32, 257, C, C) if C == 4 else (5, C, C)
    a, na = _rand(rng, shape)
    na = na + 2 * np.eye(C)
    a = ComplexTensor(torch.from_numpy(na.real.astype(np.float32)), torch.from_numpy(na.imag.astype(np.float32)))
    inv = _dev(a, device).inverse()
    _same(inv, np.linalg.inv(na.astype(np.complex128)), tol=5e-5)
    # A A^-1 = I through the matmul kernel
    eye = _dev(a, device) @ inv
    want = np.broadcast_to(np.eye(C, dtype=np.complex128), na.shape)
    assert np.abs(eye.real.cpu().numpy() + 1j * eye.imag.cpu().numpy() - want).max() < 5e-4


def test_larger_operands_keep_working(device):
    """beyond the kernels' range (K > 64, C > 8) the torch forms answer"""
    rng = np.random.default_rng(9)
    a, na = _rand(rng, (3, 10, 100))
    b, nb = _rand(rng, (3, 100, 6))
    _same(_dev(a, device) @ _dev(b, device), na @ nb, tol=5e-5)
    m, nm = _rand(rng, (2, 12, 12))
    nm = nm + 4 * np.eye(12)
    m = ComplexTensor(torch.from_numpy(nm.real.astype(np.float32)), torch.from_numpy(nm.imag.astype(np.float32)))
    _same(_dev(m, device).inverse(), np.linalg.inv(nm.astype(np.complex128)), tol=2e-4)
