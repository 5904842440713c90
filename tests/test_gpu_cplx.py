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


@pytest.mark.parametrize("scale", [1e-21, 3e19])
def test_inverse_of_matrices_whose_squared_pivots_leave_the_fp32_range(device, scale):
    """th.inverse of the real embedding (aps/cplx.py:268-278) inverts a matrix of entries ~1e-21 or ~3e19 without
    complaint; the elimination forms 1 / pivot from the pivot scaled by its larger part (|pivot|^2 would be 0 / inf)
    and only a pivot that IS zero or non-finite counts as singular"""
    import aps_amd.cplx as cplx_mod
    rng = np.random.default_rng(23)
    _, na = _rand(rng, (7, 4, 4))
    na = (na + 2 * np.eye(4)) * scale
    a = ComplexTensor(torch.from_numpy(na.real.astype(np.float32)), torch.from_numpy(na.imag.astype(np.float32)))
    inv = _dev(a, device).inverse()
    want = np.linalg.inv(na.astype(np.complex128))
    got = inv.real.cpu().numpy().astype(np.float64) + 1j * inv.imag.cpu().numpy()
    assert np.abs(got - want).max() < 5e-5 * np.abs(want).max()
    assert cplx_mod.singular_matrices(device) == 0


def test_inverse_of_a_singular_matrix_raises_like_th_inverse(device):
    """aps/cplx.py:268-278 -> th.inverse raises on a singular matrix; aps_cplx_inverse counts the matrices whose
    elimination meets a zero (or non-finite) pivot and the wrapper raises torch.linalg.LinAlgError -- after the
    error the next, regular call works again (the counter is reset)"""
    import aps_amd.cplx as cplx_mod
    rng = np.random.default_rng(17)
    a, na = _rand(rng, (6, 4, 4))
    a.real[2].zero_()
    a.imag[2].zero_()           # one exactly singular matrix of the batch
    with pytest.raises(torch.linalg.LinAlgError):
        _dev(a, device).inverse()
    with pytest.raises(RuntimeError):   # (the reference: th.inverse of the real embedding, the same class)
        top = torch.cat([a.real, -a.imag], -1)
        torch.linalg.inv(torch.cat([top, torch.cat([a.imag, a.real], -1)], -2))
    b, nb = _rand(rng, (6, 4, 4))
    nb = nb + 2 * np.eye(4)
    b = ComplexTensor(b.real + 2 * torch.eye(4), b.imag)
    _same(_dev(b, device).inverse(), np.linalg.inv(nb.astype(np.complex128)))
    assert cplx_mod.singular_matrices(device) == 0
    # the deferred policy: the call does not look (no host stall per inverse), the count is read by the caller
    cplx_mod.INVERSE_SINGULAR_POLICY = "deferred"
    try:
        _dev(a, device).inverse()
        _dev(a, device).inverse()
        assert cplx_mod.singular_matrices(device) == 2
        assert cplx_mod.singular_matrices(device) == 0
    finally:
        cplx_mod.INVERSE_SINGULAR_POLICY = "sync"


@pytest.mark.parametrize("cond", [1e6, 1e7, 1e8])
def test_inverse_eval_and_training_share_the_arithmetic_on_ill_conditioned_covariances(device, cond):
    """Rounds 1-4 inverted with the Gauss-Jordan kernel in eval and with torch's LU of the 2C x 2C embedding under
    autograd.  Now both modes run aps_cplx_inverse (autograd: + the adjoint -Y^H G Y^H on aps_cplx_matmul): the
    two inverses are the SAME bits on Hermitian positive definite 4 x 4 matrices of condition number 1e6 ... 1e8,
    both within cond x 2^-20 of the float64 inverse (relative to its norm), and the gradient of a real loss
    matches torch's own through float64 complex linalg"""
    g = torch.Generator().manual_seed(int(np.log10(cond)))
    B, C = 64, 4
    q, _ = torch.linalg.qr(torch.randn(B, C, C, dtype=torch.complex128, generator=g))
    ev = torch.logspace(0, -np.log10(cond), C, dtype=torch.float64)[None].expand(B, C)
    m = (q * ev[:, None, :].to(torch.complex128)) @ q.conj().transpose(-1, -2)
    re, im = m.real.float().to(device), m.imag.float().to(device)
    with torch.no_grad():
        y_eval = ComplexTensor(re, im).inverse()
    re_g, im_g = re.clone().requires_grad_(True), im.clone().requires_grad_(True)
    y_train = ComplexTensor(re_g, im_g).inverse()
    assert torch.equal(y_eval.real, y_train.real) and torch.equal(y_eval.imag, y_train.imag)
    m32 = torch.complex(re.double().cpu(), im.double().cpu())   # (the fp32-rounded matrices the kernels saw)
    want = torch.linalg.inv(m32)
    got = torch.complex(y_eval.real.double().cpu(), y_eval.imag.double().cpu())
    err = ((got - want).abs().amax((-1, -2)) / want.abs().amax((-1, -2))).max().item()
    assert err <= cond * 2.0 ** -20, (cond, err)
    # gradient of L = sum Re(W . Y) + Im-part mix, against float64 autograd
    w_re = torch.randn(B, C, C, generator=g).to(device)
    w_im = torch.randn(B, C, C, generator=g).to(device)
    ((y_train.real * w_re).sum() + (y_train.imag * w_im).sum()).backward()
    a64_re = re.double().cpu().requires_grad_(True)
    a64_im = im.double().cpu().requires_grad_(True)
    y64 = torch.linalg.inv(torch.complex(a64_re, a64_im))
    ((y64.real * w_re.double().cpu()).sum() + (y64.imag * w_im.double().cpu()).sum()).backward()
    scale = max(a64_re.grad.abs().max().item(), a64_im.grad.abs().max().item())
    gerr = max((re_g.grad.double().cpu() - a64_re.grad).abs().max().item(),
               (im_g.grad.double().cpu() - a64_im.grad).abs().max().item()) / scale
    assert gerr <= cond * 2.0 ** -19, (cond, gerr)


def test_mvdr_solve_counts_singular_systems(device):
    """the MVDR solve kernels (fused tail, stand-alone weight kernel) count the (n, f) whose Rn + eps I has a
    zero / non-finite pivot -- where the reference's Rn.inverse() raises (mvdr.py:89-92) -- and MvdrBeamformer
    raises torch.linalg.LinAlgError under singular_policy = "sync"; regular inputs count nothing"""
    from aps_amd import ops
    from aps_amd.asr.filter import mvdr as M
    torch.manual_seed(5)
    N, C, F = 3, 4, 257
    mv = M.MvdrBeamformer(F, att_dim=64).to(device)
    g = torch.Generator().manual_seed(6)
    a = torch.randn(N, F, C, C, 2, generator=g)
    herm = torch.stack([a[..., 0] + a[..., 0].transpose(-1, -2), a[..., 1] - a[..., 1].transpose(-1, -2)], -1)
    cov_s = herm.to(device)
    cov_n = (herm + torch.stack([4 * torch.eye(C), torch.zeros(C, C)], -1)).to(device)
    u = torch.softmax(torch.randn(N, C, generator=g), -1).to(device)
    ops.MVDR_SINGULAR.count()
    mv.singular_policy = "sync"
    w = mv.derive_weight(cov_s, cov_n, u)
    assert torch.isfinite(w).all() and ops.MVDR_SINGULAR.count() == 0
    bad = cov_n.clone()
    bad[1, 7] = float("nan")
    bad[2, 100] = -1e-5 * torch.stack([torch.eye(C), torch.zeros(C, C)], -1).to(device)   # Rn + eps I = 0 exactly
    with pytest.raises(torch.linalg.LinAlgError):
        mv.derive_weight(cov_s, bad, u, eps=1e-5)
    mv.singular_policy = "manual"
    mv.derive_weight(cov_s, bad, u, eps=1e-5)
    assert ops.MVDR_SINGULAR.count() == 2
