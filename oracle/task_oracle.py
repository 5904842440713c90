"""
TEST INFRASTRUCTURE -- CPU restatement of the task-side consumers of the STFT / covariance kernels
(SURVEY.md 8f row 4): the spectral-approximation objectives of aps/task/sse.py:207-455 (FreqSaTask,
LinearFreqSaTask, MelFreqSaTask) with the permutation objective of aps/task/objf.py:238-369, and the
maximum-likelihood objective of aps/task/ml.py:14-122 (MlEnhTask).  Plain torch ops in the
reference's order, differentiable (the GPU tests compare gradients through it).  Pinned by
tests/golden/task_*.npz, recorded from the reference's own task classes (tests/test_oracle_tasks.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
"""
from itertools import permutations

import torch
import torch.nn.functional as F

from oracle import aps_oracle as ao

EPSILON = ao.EPSILON


# ---- aps/task/objf.py:238-369 ------------------------------------------------------------------
def multiple_objf(inp, ref, objf, weight=None, transform=None):
    if weight is None:
        weight = [1 / len(inp)] * len(inp)
    if transform:
        inp, ref = [transform(i) for i in inp], [transform(r) for r in ref]
    return sum(w * objf(o, r) for w, o, r in zip(weight, inp, ref))


def permu_invariant_objf(inp, ref, objf, transform=None):
    if transform:
        inp, ref = [transform(i) for i in inp], [transform(r) for r in ref]
    if len(inp) == 1:
        return objf(inp[0], ref[0])
    mat = torch.stack([sum(objf(inp[s], ref[t]) for s, t in enumerate(p)) / len(p)
                       for p in permutations(range(len(inp)))])
    return torch.min(mat, dim=0)[0]


def hybrid_permu_objf(out, ref, objf, transform=None, weight=None, permute=True, permu_num_spks=2):
    if not permute:
        return multiple_objf(out, ref, objf, weight=weight, transform=transform)
    loss = permu_invariant_objf(out[:permu_num_spks], ref[:permu_num_spks], objf, transform)
    if len(out) > permu_num_spks:
        num_weight = len(out) - (permu_num_spks - 1)
        if weight is None:
            weight = [1 / num_weight] * num_weight
        loss = weight[0] * loss + multiple_objf(out[permu_num_spks:], ref[permu_num_spks:], objf,
                                                weight=weight[1:])
    return loss


# ---- aps/task/sse.py:207-455 -------------------------------------------------------------------
def polar_stft(wav, frame_len=512, frame_hop=256, window="sqrthann"):
    """ctx("forward_stft")(wav, return_polar=True): N x F x T x (mag, phase)"""
    return ao.stft(wav, frame_len, frame_hop, window, polar=True)


def ref_magnitude(mix_polar, ref_polar, phase_sensitive=False, truncated=-1):
    mag, pha = ref_polar[..., 0], ref_polar[..., 1]
    if phase_sensitive:
        mag = mag * torch.clamp(torch.cos(pha - mix_polar[..., 1]), min=0)
    if truncated > 0:
        mag = torch.min(mag, truncated * mix_polar[..., 0])
    return mag


def freq_sa_loss(masks, mix, refs, objf="L2", phase_sensitive=False, truncated=-1, permute=True,
                 masking=True, num_spks=2, weight=None, mel=None, mel_log=False, power_mag=False,
                 stft_kwargs=None):
    """LinearFreqSaTask / MelFreqSaTask .forward(egs)["loss"] for a network that emitted `masks`
    (list of N x F x T); mel = the [M, F] matrix of MelFreqSaTask (already scaled) or None"""
    kw = stft_kwargs or {}
    mix_polar = polar_stft(mix[:, 0] if mix.dim() == 3 else mix, **kw)
    ref_polar = [polar_stft(r, **kw) for r in refs]
    targets = [ref_magnitude(mix_polar, r, phase_sensitive, truncated) for r in ref_polar]
    out = [m * mix_polar[..., 0] for m in masks] if masking else list(masks)
    if mel is None:
        fn = F.l1_loss if objf == "L1" else F.mse_loss

        def pair(o, r):
            return fn(o, r, reduction="none").mean(-1).sum(-1)

        transform = None
    else:
        def pair(o, r):
            return F.mse_loss(o, r, reduction="none").mean(-1).sum(-1)

        def transform(t):
            if power_mag:
                t = t**2
            m = F.conv1d(t, mel[..., None])
            return torch.log(1 + m) if mel_log else m

    weight = None if weight is None else list(map(float, weight.split(",")))
    loss = hybrid_permu_objf(out, targets, pair, transform=transform, weight=weight,
                             permute=permute, permu_num_spks=num_spks)
    return loss.mean()


# ---- aps/task/ml.py:14-122 ---------------------------------------------------------------------
def hermitian_det(br, bi, eps=EPSILON):
    m = torch.cat([br, -bi], -1)
    n = torch.cat([bi, br], -1)
    ev, _ = torch.linalg.eigh(torch.cat([m, n], -2), UPLO="U")
    return torch.clamp(torch.cumprod(ev[..., ::2], dim=-1)[..., -1], min=eps)


def ml_covar(mask, xr, xi, eps=EPSILON):
    """mask N x F x T, obs N x F x C x T -> Hermitian B (re, im) N x F x C x C (ml.py:38-61)"""
    C = xr.shape[-2]
    m = mask.unsqueeze(-2)
    ar, ai = xr * m, xi * m
    br, bi = xr.transpose(-1, -2), -1.0 * xi.transpose(-1, -2)
    rr = torch.matmul(ar, br) - torch.matmul(ai, bi)
    ri = torch.matmul(ai, br) + torch.matmul(ar, bi)
    den = torch.clamp(m.sum(-1, keepdim=True), min=eps)
    rr, ri = C * rr / den, C * ri / den
    return (rr + rr.transpose(-1, -2)) / 2, (ri - ri.transpose(-1, -2)) / 2


def ml_log_pdf(mask, xr, xi, eps=EPSILON):
    """MlEnhTask.log_pdf: mask N x F x T, obs N x F x C x T -> N x F x T"""
    C = xr.shape[-2]
    br, bi = ml_covar(mask, xr, xi, eps)
    br = br + torch.eye(C, dtype=br.dtype) * eps
    det = hermitian_det(br, bi, eps)
    ir, ii = ao.cplx_inverse(br, bi)
    yr = torch.matmul(ir, xr) - torch.matmul(ii, xi)  # B^-1 obs: N x F x C x T
    yi = torch.matmul(ii, xr) + torch.matmul(ir, xi)
    k = (xr * yr + xi * yi).sum(-2)  # Re conj(obs) . (B^-1 obs)
    k = torch.clamp(k, min=eps)
    return -C * torch.log(k) - torch.log(det[..., None])


def ml_loss(ms, xr, xi, eps=EPSILON):
    """MlEnhTask.forward given what the network returned: obs N x C x F x T, ms N x T x F"""
    xr, xi = xr.transpose(1, 2), xi.transpose(1, 2)
    ms = ms.transpose(-1, -2)
    ps, pn = ml_log_pdf(ms, xr, xi, eps), ml_log_pdf(1 - ms, xr, xi, eps)
    return -torch.mean(torch.log((torch.exp(ps) + torch.exp(pn)) * 0.5))
