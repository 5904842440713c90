"""
The backward arithmetic of the HIP build, checked on the CPU: tests/csrc/grad_host.cc compiles the
index functors of aps_amd/csrc/grad_core.h and the C-ABI marshalling of aps_amd/csrc/grad_api.inc
for the host (g++), and every `host_*` entry point is compared here with torch autograd through the
CPU oracle (oracle/aps_oracle.py: the restatement of the reference's arithmetic) or through the
torch layer the reference itself uses (nn.LSTM, nn.LayerNorm, nn.BatchNorm1d, F.conv1d).  The GPU
build runs the SAME functors one thread per index (tests/test_gpu_backward.py checks that end).
"""
import ctypes as C
import os
import subprocess

import pytest
import torch
import torch.nn.functional as F

from tests.conftest import ROOT
from oracle import aps_oracle as ao


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    from aps_amd import _native
    lib_path = str(tmp_path_factory.mktemp("grad_host") / "libgrad_host.so")
    src = os.path.join(ROOT, "tests", "csrc", "grad_host.cc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", lib_path, src],
                   check=True)
    lib = C.CDLL(lib_path)
    inc = open(os.path.join(ROOT, "aps_amd", "csrc", "grad_api.inc")).read()
    for name, (res, args) in _native.SIGNATURES.items():
        host_name = "host_" + name[4:]
        if f"APS_GRAD_API({name[4:]})" in inc or name == "aps_lstm_backward_sweep":
            fn = getattr(lib, host_name)
            fn.restype, fn.argtypes = res, args
    return lib


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def close(got, want, tol=2e-5, what=""):
    scale = want.abs().max().clamp_min(1e-20)
    err = ((got.double() - want.double()).abs().max() / scale).item()
    assert err <= tol, f"{what}: scaled error {err:.3e} > {tol:.1e}"


ACTS = {0: lambda x: x, 1: torch.relu, 2: F.silu, 3: torch.sigmoid, 4: torch.tanh, 5: F.gelu,
        6: F.leaky_relu,  # nn.LeakyReLU() of the DCCRN blocks (slope 0.01)
        7: torch.square}  # PowerTransform(2) inside a differentiable feature chain


@pytest.mark.parametrize("act", sorted(ACTS))
def test_activation_forward_backward(host, act):
    torch.manual_seed(act)
    pre = torch.randn(1000, requires_grad=True)
    res, g = torch.randn(1000), torch.randn(1000)
    want = ACTS[act](pre) * 0.5 + res
    want.backward(g)
    out, gp = torch.empty(1000), torch.empty(1000)
    assert host.host_act_forward(P(pre.detach()), P(res), P(out), 1000, act, 0.5, None) == 0
    assert host.host_act_backward(P(g), P(pre.detach()), P(gp), 1000, act, 0.5, None) == 0
    close(out, want.detach(), what="act forward")
    close(gp, pre.grad, what="act backward")


def test_bias_add_and_embedding_rows(host):
    torch.manual_seed(12)
    x, b = torch.randn(7, 5), torch.randn(5)
    out = torch.empty(7, 5)
    assert host.host_row_bias_add(P(x), P(b), P(out), 7, 5, None) == 0
    assert torch.equal(out, x + b)
    emb = torch.nn.Embedding(9, 6)
    index = torch.tensor([0, 0, 1, 4, 8, 8, 8, 3])
    g = torch.randn(8, 6)
    emb(index).backward(g)
    g_w = torch.empty(9, 6)
    assert host.host_gather_rows_backward(P(index), P(g), P(g_w), 8, 9, 6, None) == 0
    close(g_w, emb.weight.grad, what="embedding rows")


def colreduce(host, mode, A, B=None, v1=None, v2=None, scale=1.0, out=None):
    rows, cols = A.shape
    ws = torch.empty(host.host_colreduce_workspace(rows, cols) // 4)
    acc = 0 if out is None else 1
    out = torch.empty(cols) if out is None else out
    rc = host.host_colreduce(mode, P(A), P(B), P(v1), P(v2), rows, cols, A.stride(0),
                             0 if B is None else B.stride(0), scale, acc, P(out), P(ws), None)
    assert rc == 0
    return out


def test_column_reductions(host):
    torch.manual_seed(0)
    A, B = torch.randn(1000, 37), torch.randn(1000, 37)
    v1, v2 = torch.randn(37), torch.rand(37)
    close(colreduce(host, 0, A), A.sum(0), what="sum")
    close(colreduce(host, 1, A, B, scale=0.5), 0.5 * (A * B).sum(0), what="dot")
    close(colreduce(host, 2, A, v1=v1), ((A - v1)**2).sum(0), what="centred squares")
    close(colreduce(host, 3, A, B, v1, v2), (A * (B - v1) * v2).sum(0), what="bn term")
    wide = torch.randn(70000, 3)  # more rows than one chunk of partials covers
    close(colreduce(host, 0, wide), wide.double().sum(0).float(), what="many chunks")
    acc = torch.ones(37)
    close(colreduce(host, 0, A, out=acc), 1 + A.sum(0), what="accumulate")
    view = torch.randn(50, 64)[:, :20]  # pitch > cols
    close(colreduce(host, 0, view), view.sum(0), what="strided")


@pytest.mark.parametrize("with_res", [False, True])
def test_layernorm_backward(host, with_res):
    torch.manual_seed(1)
    rows, D = 23, 96
    x = torch.randn(rows, D, requires_grad=True)
    res = torch.randn(rows, D, requires_grad=True) if with_res else None
    ln = torch.nn.LayerNorm(D)
    ln.weight.data.uniform_(0.5, 1.5)
    ln.bias.data.normal_()
    g = torch.randn(rows, D)
    ln(x + res if with_res else x).backward(g)
    gx, t = torch.empty(rows, D), torch.empty(rows, D)
    rc = host.host_layernorm_backward(P(x.detach()), P(None if res is None else res.detach()),
                                      P(ln.weight.detach()), P(g), P(gx), P(t), rows, D, ln.eps,
                                      None)
    assert rc == 0
    close(gx, x.grad, what="g_x")
    if with_res:
        close(gx, res.grad, what="g_residual")
    close(t.sum(0), ln.weight.grad, what="g_gamma")
    close(g.sum(0), ln.bias.grad, what="g_beta")


def test_batchnorm_training(host):
    torch.manual_seed(2)
    rows, D = 300, 24
    x = (torch.randn(rows, D) * 2 + 1).requires_grad_(True)
    bn = torch.nn.BatchNorm1d(D)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_()
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    g = torch.randn(rows, D)
    y = bn(x)
    y.backward(g)
    mean, rstd = torch.empty(D), torch.empty(D)
    ws = torch.empty(host.host_batchnorm_workspace(rows, D) // 4)
    xd = x.detach()
    assert host.host_batchnorm_stats(P(xd), rows, D, bn.eps, bn.momentum, P(mean), P(rstd), P(rm),
                                     P(rv), P(ws), None) == 0
    close(rm, bn.running_mean, what="running mean")
    close(rv, bn.running_var, what="running var")
    out = torch.empty(rows, D)
    assert host.host_batchnorm_apply(P(xd), P(mean), P(rstd), P(bn.weight.detach()),
                                     P(bn.bias.detach()), P(out), rows, D, None) == 0
    close(out, y.detach(), what="bn forward")
    s1 = colreduce(host, 0, g)
    s2 = colreduce(host, 3, g, xd, mean, rstd)
    close(s1, bn.bias.grad, what="g_beta")
    close(s2, bn.weight.grad, what="g_gamma")
    gx = torch.empty(rows, D)
    assert host.host_batchnorm_backward(P(xd), P(mean), P(rstd), P(bn.weight.detach()), P(g),
                                        P(s1), P(s2), P(gx), rows, D, None) == 0
    close(gx, x.grad, what="g_x (batch statistics)")
    # eval mode: the statistics are constants
    bn.eval()
    x2 = xd.clone().requires_grad_(True)
    bn(x2).backward(g)
    ev_mean, ev_rstd = bn.running_mean.clone(), (bn.running_var + bn.eps).rsqrt()
    assert host.host_batchnorm_backward(P(xd), P(ev_mean), P(ev_rstd), P(bn.weight.detach()), P(g),
                                        None, None, P(gx), rows, D, None) == 0
    close(gx, x2.grad, what="g_x (running statistics)")


def test_softmax_rows(host):
    torch.manual_seed(3)
    x = torch.randn(11, 4, requires_grad=True)
    g = torch.randn(11, 4)
    y = torch.softmax(x, -1)
    y.backward(g)
    out, gx = torch.empty(11, 4), torch.empty(11, 4)
    assert host.host_softmax_rows(P(x.detach()), P(out), 11, 4, None) == 0
    assert host.host_softmax_rows_backward(P(out), P(g), P(gx), 11, 4, None) == 0
    close(out, y.detach(), what="softmax")
    close(gx, x.grad, what="softmax backward")


@pytest.mark.parametrize("norm_mean,norm_var", [(True, True), (True, False), (False, True)])
def test_abs_mel_log_cmvn_backward(host, norm_mean, norm_var):
    """AsrTransform("abs-mel-log-cmvn") on the beamformer output (asr.py:306-332, 360-464, 576-618)
    = magnitude -> mel GEMM -> log + CMVN rows; the two functors around the GEMM vs autograd
    through the oracle"""
    torch.manual_seed(4)
    rows, Fb, M = 17, 33, 12
    z = torch.randn(rows, Fb, 2, requires_grad=True)
    mel = torch.rand(M, Fb)
    mag = ((z[..., 0] + ao.EPSILON)**2 + z[..., 1]**2).sqrt()
    m = F.linear(mag, mel)
    m.retain_grad()
    mag.retain_grad()
    out = ao.cmvn(ao.log_feature(m), norm_mean=norm_mean, norm_var=norm_var)
    g = torch.randn(rows, M)
    out.backward(g)
    gm = torch.empty(rows, M)
    rc = host.host_log_cmvn_backward(P(m.detach()), P(g), P(gm), rows, M, 1, int(norm_mean),
                                     int(norm_var), ao.EPSILON, 0.0, ao.EPSILON, None)
    assert rc == 0
    close(gm, m.grad, what="log + cmvn backward")
    mag_out = torch.empty(rows, Fb)
    assert host.host_magnitude_forward(P(z.detach()), P(mag_out), rows * Fb, ao.EPSILON, None) == 0
    close(mag_out, mag.detach(), what="magnitude forward")
    gz = torch.empty(rows, Fb, 2)
    g_mag = mag.grad.contiguous()
    rc = host.host_magnitude_backward(P(z.detach()), P(g_mag), P(gz), rows * Fb,
                                      ao.EPSILON, None)
    assert rc == 0
    close(gz, z.grad, what="magnitude backward")


def test_glu_dwconv_backward(host):
    torch.manual_seed(5)
    N, T, D, K = 3, 19, 8, 5
    x = torch.randn(N, T, 2 * D, requires_grad=True)
    w = torch.randn(D, K, requires_grad=True)
    b = torch.randn(D, requires_grad=True)
    glu = F.glu(x, dim=-1)  # N x T x D
    c = F.conv1d(glu.transpose(1, 2), w[:, None, :], b, padding=(K - 1) // 2, groups=D)
    c = c.transpose(1, 2)
    g = torch.randn(N, T, D)
    c.backward(g)
    gx, gw = torch.empty(N, T, 2 * D), torch.empty(D, K)
    ws = torch.empty(host.host_glu_dwconv_backward_workspace(N, T, D, K) // 4)
    rc = host.host_glu_dwconv_backward(P(x.detach()), P(w.detach()), P(g), P(gx), P(gw), N, T, D, K,
                                       P(ws), None)
    assert rc == 0
    close(gx, x.grad, what="g_x")
    close(gw, w.grad, what="g_w")
    close(colreduce(host, 0, g.reshape(-1, D)), b.grad, what="g_bias")


@pytest.mark.parametrize("T,with_pad", [(19, True), (3, True), (19, False)])
def test_glu_dwconv_backward_causal(host, T, with_pad):
    """the causal form: K - 1 frames in front of the sequence that carry glu(pad_bias) (the conformer
    pads the projected sequence's INPUT with zeros, so the padded frames see only the bias), T < K too"""
    torch.manual_seed(15)
    N, D, K = 3, 8, 5
    x = torch.randn(N, T, 2 * D, requires_grad=True)
    w = torch.randn(D, K, requires_grad=True)
    pb = torch.randn(2 * D, requires_grad=True)
    front = pb[None, None, :].expand(N, K - 1, 2 * D) if with_pad else torch.zeros(N, K - 1, 2 * D)
    full = torch.cat([front, x], 1)
    glu = F.glu(full, dim=-1)
    if not with_pad:  # zeros in front of the GLU OUTPUT (glu(0) = 0 anyway)
        assert float(glu.detach()[:, :K - 1].abs().max()) == 0
    c = F.conv1d(glu.transpose(1, 2), w[:, None, :], None, groups=D).transpose(1, 2)
    assert c.shape == (N, T, D)
    g = torch.randn(N, T, D)
    c.backward(g)
    gx, gw, gp = torch.empty(N, T, 2 * D), torch.empty(D, K), torch.empty(2 * D)
    ws = torch.empty(host.host_glu_dwconv_backward_workspace(N, T, D, K) // 4)
    rc = host.host_glu_dwconv_backward_causal(
        P(x.detach()), P(w.detach()), P(g), P(pb.detach()) if with_pad else None, P(gx), P(gw),
        P(gp) if with_pad else None, N, T, D, K, P(ws), None)
    assert rc == 0
    close(gx, x.grad, what="g_x")
    close(gw, w.grad, what="g_w")
    if with_pad:
        close(gp, pb.grad, what="g_pad_bias")


def test_im2col_gives_the_conv_weight_gradient(host):
    torch.manual_seed(6)
    N, H, W, Ci, Co, K, s, p = 2, 9, 11, 3, 5, 3, 2, 1
    x = torch.randn(N, Ci, H, W)
    w = torch.randn(Co, Ci, K, K, requires_grad=True)
    y = F.conv2d(x, w, stride=s, padding=p)
    g = torch.randn_like(y)
    y.backward(g)
    Ho, Wo = y.shape[-2:]
    ld = 28  # KH KW Ci = 27 padded to a multiple of 4
    patches = torch.empty(N * Ho * Wo, ld)
    xl = x.permute(0, 2, 3, 1).contiguous()
    rc = host.host_im2col_nhwc(P(xl), P(patches), N, H, W, Ci, K, K, s, s, p, p, Ho, Wo, ld, None)
    assert rc == 0
    assert patches[:, 27:].abs().max() == 0
    gy = g.permute(0, 2, 3, 1).reshape(-1, Co)  # [M, Co] channels-last
    gw = gy.t() @ patches[:, :27]  # [Co, (kh, kw, ci)]
    close(gw.view(Co, K, K, Ci), w.grad.permute(0, 2, 3, 1), what="g_W through im2col")
    # and the forward itself through the same patches
    yl = patches[:, :27] @ w.detach().permute(0, 2, 3, 1).reshape(Co, -1).t()
    close(yl, y.detach().permute(0, 2, 3, 1).reshape(-1, Co), what="conv as patches x W")


def rel_attention_reference(qkv, lens, rel, zero, H):
    """softmax((q k^T + q rel[j - i + zero]) / sqrt(dh)) v with key masks (impl.py:225-296)"""
    N, T, _ = qkv.shape
    q, k, v = qkv.view(N, T, 3, H, -1).unbind(2)  # N x T x H x dh
    dh = q.shape[-1]
    s = torch.einsum("nihd,njhd->nhij", q, k)
    if rel is not None:
        idx = torch.arange(T)[None, :] - torch.arange(T)[:, None] + zero  # [i, j] -> j - i + zero
        ok = (idx >= 0) & (idx < rel.shape[0])
        table = rel[idx.clamp(0, rel.shape[0] - 1)] * ok[..., None]  # T x T x dh
        s = s + torch.einsum("nihd,ijd->nhij", q, table)
    s = s / dh**0.5
    if lens is not None:
        mask = torch.arange(T)[None, :] >= lens[:, None]
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1)
    return torch.einsum("nhij,njhd->nihd", p, v).reshape(N, T, -1)


@pytest.mark.parametrize("dh", [8, 32, 64])  # 8: the generic recomputing form; 32 / 64: registers
@pytest.mark.parametrize("use_rel,use_lens", [(False, False), (True, False), (True, True)])
def test_attention_backward(host, use_rel, use_lens, dh):
    torch.manual_seed(7)
    N, T, H = 2, 9, 3
    qkv = torch.randn(N, T, 3 * H * dh, requires_grad=True)
    lens = torch.tensor([9, 6]) if use_lens else None
    R = 2 * T - 1 - 4  # a table shorter than 2T - 1: rows outside count as zero
    rel = torch.randn(R, dh, requires_grad=True) if use_rel else None
    zero = (R - 1) // 2
    ctx = rel_attention_reference(qkv, lens, rel, zero, H)
    g = torch.randn_like(ctx)
    ctx.backward(g)
    g_qkv = torch.empty(N, T, 3 * H * dh)
    part = torch.empty(N * H, R, dh) if use_rel else None
    ws = torch.empty(host.host_attention_backward_workspace(N, T, H) // 4)
    rc = host.host_attention_backward(P(qkv.detach()), P(lens), P(None if rel is None else
                                                                 rel.detach()), zero,
                                      R if use_rel else 0, P(g), P(g_qkv), P(part), N, T, H, dh,
                                      0.0, 0, P(ws), None)
    assert rc == 0
    close(g_qkv, qkv.grad, what="g_qkv")
    if use_rel:
        close(part.sum(0), rel.grad, what="g_rel")


@pytest.mark.parametrize("use_lens", [False, True])
def test_lstm_backward_through_time(host, use_lens):
    torch.manual_seed(8)
    N, T, D, H = 3, 7, 5, 8
    rnn = torch.nn.LSTM(D, H, batch_first=True)
    x = torch.randn(N, T, D, requires_grad=True)
    lens = torch.tensor([7, 4, 6]) if use_lens else None
    if use_lens:
        packed = torch.nn.utils.rnn.pack_padded_sequence(x, lens.tolist(), batch_first=True,
                                                         enforce_sorted=False)
        y, _ = torch.nn.utils.rnn.pad_packed_sequence(rnn(packed)[0], batch_first=True,
                                                      total_length=T)
    else:
        y, _ = rnn(x)
    g = torch.randn(N, T, H)
    y.backward(g)
    w_ih, w_hh = rnn.weight_ih_l0.detach(), rnn.weight_hh_l0.detach()
    b_ih, b_hh = rnn.bias_ih_l0.detach(), rnn.bias_hh_l0.detach()
    yd = y.detach().contiguous()
    hprev = torch.empty(N, T, H)
    assert host.host_time_shift(P(yd), P(hprev), N, T, H, None) == 0
    assert torch.equal(hprev[:, 1:], yd[:, :-1]) and hprev[:, 0].abs().max() == 0
    pre = F.linear(x.detach(), w_ih, b_ih).contiguous()
    hh = F.linear(hprev, w_hh).contiguous()
    gates, c = torch.empty(N, T, 4 * H), torch.empty(N, T, H)
    assert host.host_lstm_gate_scan(P(pre), P(hh), P(b_hh), P(lens), P(gates), P(c), N, T, H,
                                    None) == 0
    # the recomputed gates reproduce the layer output: h = o tanh(c)
    close(gates[..., 3 * H:] * torch.tanh(c), yd, what="recomputed h")
    g_pre, g_h, g_c = torch.empty(N, T, 4 * H), torch.empty(N, H), torch.empty(N, H)
    w_hh_t = w_hh.t().contiguous()  # (kept alive across the call: P() only takes the address)
    rc = host.host_lstm_backward_sweep(P(gates), P(c), P(g), P(w_hh_t), P(lens),
                                       P(g_pre), P(g_h), P(g_c), N, T, H, None)
    assert rc == 0
    flat = g_pre.view(-1, 4 * H)
    close(g_pre @ w_ih, x.grad, what="g_x")
    close(flat.t() @ x.detach().view(-1, D), rnn.weight_ih_l0.grad, what="g_W_ih")
    close(flat.t() @ hprev.view(-1, H), rnn.weight_hh_l0.grad, what="g_W_hh")
    close(flat.sum(0), rnn.bias_ih_l0.grad, what="g_b_ih")
    close(flat.sum(0), rnn.bias_hh_l0.grad, what="g_b_hh")


def mvdr_inputs(N=2, C=4, Fb=9, T=13, seed=9):
    g = torch.Generator().manual_seed(seed)
    xr, xi = torch.randn(N, C, Fb, T, generator=g), torch.randn(N, C, Fb, T, generator=g)
    store = torch.stack([xr, xi], -1).permute(0, 1, 3, 2, 4).contiguous()  # N x C x T x F x 2
    return xr, xi, store, g


@pytest.mark.parametrize("mask_norm,use_lens", [(True, False), (True, True), (False, False)])
def test_covariance_backward(host, mask_norm, use_lens):
    N, C, Fb, T = 2, 4, 9, 13
    xr, xi, store, gen = mvdr_inputs(N, C, Fb, T)
    mask = torch.rand(N, T, Fb, generator=gen).requires_grad_(True)
    lens = torch.tensor([13, 9]) if use_lens else None
    pm = ao.process_mask(mask, lens, mask_norm)  # N x F x T
    rr, ri = ao.covar(pm, xr, xi)
    gr, gi = torch.randn(N, Fb, C, C, generator=gen), torch.randn(N, Fb, C, C, generator=gen)
    (rr * gr + ri * gi).sum().backward()
    cov = torch.stack([rr.detach(), ri.detach()], -1).contiguous()
    g_cov = torch.stack([gr, gi], -1).contiguous()
    g_mask = torch.empty(N, T, Fb)
    rc = host.host_mvdr_covariance_backward(P(store), P(mask.detach()), P(lens), P(cov), P(g_cov),
                                            P(g_mask), N, C, T, Fb, store.stride(0),
                                            store.stride(1), store.stride(2), int(mask_norm), None, None)
    assert rc == 0
    close(g_mask, mask.grad, what="g_mask")


@pytest.mark.parametrize("mask_norm,use_lens", [(True, False), (True, True), (False, True)])
def test_covariance_backward_implicit_noise_mask(host, mask_norm, use_lens):
    """MvdrBeamformer.forward without a noise mask (mvdr.py:131-135): Rn = estimate_covar(1 - m', X) with
    m' the processed speech mask -- Rn's gradient w.r.t. the complement (all T frames, not processed again)
    enters the speech branch's adjoint through g_sub; the sum against autograd through the oracle"""
    N, C, Fb, T = 2, 4, 9, 13
    xr, xi, store, gen = mvdr_inputs(N, C, Fb, T, seed=5)
    mask = torch.rand(N, T, Fb, generator=gen).requires_grad_(True)
    lens = torch.tensor([13, 9]) if use_lens else None
    pm = ao.process_mask(mask, lens, mask_norm)  # N x F x T
    rs_r, rs_i = ao.covar(pm, xr, xi)
    rn_r, rn_i = ao.covar(1 - pm, xr, xi)
    g = [torch.randn(N, Fb, C, C, generator=gen) for _ in range(4)]
    (rs_r * g[0] + rs_i * g[1] + rn_r * g[2] + rn_i * g[3]).sum().backward()
    comp = (1 - pm).detach().transpose(1, 2).contiguous()  # N x T x F
    cov_n = torch.stack([rn_r.detach(), rn_i.detach()], -1).contiguous()
    g_cov_n = torch.stack([g[2], g[3]], -1).contiguous()   # (named: P() takes an address, not a reference)
    g_comp = torch.empty(N, T, Fb)
    rc = host.host_mvdr_covariance_backward(P(store), P(comp), None, P(cov_n), P(g_cov_n), P(g_comp), N, C, T,
                                            Fb, store.stride(0), store.stride(1), store.stride(2), 0, None,
                                            None)
    assert rc == 0
    cov_s = torch.stack([rs_r.detach(), rs_i.detach()], -1).contiguous()
    g_cov_s = torch.stack([g[0], g[1]], -1).contiguous()
    g_mask = torch.empty(N, T, Fb)
    raw = mask.detach()
    rc = host.host_mvdr_covariance_backward(P(store), P(raw), P(lens), P(cov_s), P(g_cov_s), P(g_mask), N, C, T,
                                            Fb, store.stride(0), store.stride(1), store.stride(2),
                                            int(mask_norm), P(g_comp), None)
    assert rc == 0
    close(g_mask, mask.grad, what="g_mask with the implicit noise mask")


@pytest.mark.parametrize("C", [2, 4, 6])
def test_weight_and_attention_backward(host, C):
    """_derive_weight + ChannelAttention (mvdr.py:75-101, 148-174): g_w -> g_Rs, g_Rn, g_u and the
    off-diagonal magnitude the attention consumes"""
    N, Fb, T = 2, 7, 40
    xr, xi, store, gen = mvdr_inputs(N, C, Fb, T, seed=10 + C)
    ms, mn = torch.rand(N, Fb, T, generator=gen), torch.rand(N, Fb, T, generator=gen)
    rs = [t.detach().requires_grad_(True) for t in ao.covar(ms, xr, xi)]
    rn = [t.detach().requires_grad_(True) for t in ao.covar(mn, xr, xi)]
    u = torch.softmax(torch.randn(N, C, generator=gen), -1).requires_grad_(True)
    wr, wi = ao.mvdr_weight(rs, rn, u)
    gwr, gwi = torch.randn(N, Fb, C, generator=gen), torch.randn(N, Fb, C, generator=gen)
    # the attention's view of Rs
    diag = torch.eye(C, dtype=torch.bool)
    mr = rs[0].masked_fill(diag, 0).sum(-1) / (C - 1)
    mi = rs[1].masked_fill(diag, 0).sum(-1) / (C - 1)
    v = (mr**2 + mi**2).sqrt().transpose(1, 2)  # N x C x F
    gv = torch.randn(N, C, Fb, generator=gen)
    ((wr * gwr + wi * gwi).sum() + (v * gv).sum()).backward()
    cov_s = torch.stack([rs[0].detach(), rs[1].detach()], -1).contiguous()
    cov_n = torch.stack([rn[0].detach(), rn[1].detach()], -1).contiguous()
    g_w = torch.stack([gwr, gwi], -1).contiguous()
    g_s, g_n = torch.empty_like(cov_s), torch.empty_like(cov_n)
    g_u = torch.empty(N, Fb, C)
    rc = host.host_mvdr_weight_backward(P(cov_s), P(cov_n), P(u.detach()), P(g_w), P(g_s), P(g_n),
                                        P(g_u), N, C, Fb, 1e-5, None)
    assert rc == 0
    v_out = torch.empty(N, C, Fb)
    assert host.host_mvdr_offdiag_abs(P(cov_s), P(v_out), N, C, Fb, None) == 0
    close(v_out, v.detach(), what="offdiag magnitude")
    gv = gv.contiguous()
    assert host.host_mvdr_offdiag_abs_backward(P(cov_s), P(gv), P(g_s), N, C, Fb, None) == 0
    close(g_s, torch.stack([rs[0].grad, rs[1].grad], -1), tol=2e-4, what="g_Rs")
    close(g_n, torch.stack([rn[0].grad, rn[1].grad], -1), tol=2e-4, what="g_Rn")
    close(g_u.sum(1), u.grad, tol=2e-4, what="g_u")


def test_beamform_backward(host):
    N, C, Fb, T = 2, 4, 9, 13
    xr, xi, store, gen = mvdr_inputs(N, C, Fb, T, seed=21)
    wr = torch.randn(N, C, Fb, generator=gen, requires_grad=True)
    wi = torch.randn(N, C, Fb, generator=gen, requires_grad=True)
    yr, yi = ao.beamform(wr, wi, xr, xi)  # N x F x T
    gr, gi = torch.randn(N, Fb, T, generator=gen), torch.randn(N, Fb, T, generator=gen)
    (yr * gr + yi * gi).sum().backward()
    g_y = torch.stack([gr, gi], -1).transpose(1, 2).contiguous()  # N x T x F x 2
    g_w = torch.empty(N, Fb, C, 2)
    rc = host.host_mvdr_beamform_backward(P(store), P(g_y), P(g_w), N, C, T, Fb, store.stride(0),
                                          store.stride(1), store.stride(2), None)
    assert rc == 0
    want = torch.stack([wr.grad, wi.grad], -1).transpose(1, 2)  # N x F x C x 2
    close(g_w, want, what="g_w")


def test_cacgmm_log_pdf_and_adjoint(host):
    """MlEnhTask.log_pdf (ml.py:66-101) on the covariance kernel's output: forward and g_cov vs
    autograd through the task oracle (eigh-based determinant, complex inverse)"""
    from oracle import task_oracle as to
    N, C, Fb, T = 2, 4, 7, 30
    xr, xi, store, gen = mvdr_inputs(N, C, Fb, T, seed=31)
    mask = torch.rand(N, Fb, T, generator=gen)
    obs_r, obs_i = xr.transpose(1, 2), xi.transpose(1, 2)  # N x F x C x T
    # R = S / den as the covariance kernel emits it (mask_norm = 0); the log-pdf's B = C R + eps I
    rr, ri = [t.detach().requires_grad_(True) for t in ao.covar(mask, xr, xi)]
    br = (C * rr + (C * rr).transpose(-1, -2)) / 2 + torch.eye(C) * ao.EPSILON
    bi = (C * ri - (C * ri).transpose(-1, -2)) / 2
    det = to.hermitian_det(br, bi)
    ir, ii = ao.cplx_inverse(br, bi)
    yr = torch.matmul(ir, obs_r) - torch.matmul(ii, obs_i)
    yi = torch.matmul(ii, obs_r) + torch.matmul(ir, obs_i)
    k = torch.clamp((obs_r * yr + obs_i * yi).sum(-2), min=ao.EPSILON)
    lp = -C * torch.log(k) - torch.log(det[..., None])  # N x F x T
    close(lp.detach(), to.ml_log_pdf(mask, obs_r, obs_i), tol=1e-5, what="restated log-pdf")
    g = torch.randn(N, Fb, T, generator=gen)
    (lp * g).sum().backward()
    cov = torch.stack([rr.detach(), ri.detach()], -1).contiguous()
    out = torch.empty(N, T, Fb)
    rc = host.host_cacgmm_log_pdf(P(store), P(cov), P(out), N, C, T, Fb, store.stride(0),
                                  store.stride(1), store.stride(2), ao.EPSILON, None)
    assert rc == 0
    close(out.transpose(1, 2), lp.detach(), tol=2e-5, what="log-pdf")
    g_t = g.transpose(1, 2).contiguous()  # N x T x F
    g_cov = torch.empty_like(cov)
    rc = host.host_cacgmm_log_pdf_backward(P(store), P(cov), P(g_t), P(g_cov), N, C, T, Fb,
                                           store.stride(0), store.stride(1), store.stride(2),
                                           ao.EPSILON, None)
    assert rc == 0
    close(g_cov, torch.stack([rr.grad, ri.grad], -1), tol=2e-4, what="g_cov")


def test_reverse_time(host):
    torch.manual_seed(13)
    x = torch.randn(3, 7, 5)
    lens = torch.tensor([7, 4, 1])
    out = torch.empty_like(x)
    assert host.host_reverse_time(P(x), P(lens), P(out), 3, 7, 5, None) == 0
    for n, l in enumerate(lens.tolist()):
        assert torch.equal(out[n, :l], x[n, :l].flip(0)) and out[n, l:].abs().max() == 0 if l < 7 \
            else torch.equal(out[n], x[n].flip(0))
    back = torch.empty_like(x)
    assert host.host_reverse_time(P(out), P(lens), P(back), 3, 7, 5, None) == 0
    mask = (torch.arange(7)[None, :] < lens[:, None])[..., None]
    assert torch.equal(back, x * mask)  # its own inverse inside the lengths
    assert host.host_reverse_time(P(x), None, P(out), 3, 7, 5, None) == 0
    assert torch.equal(out, x.flip(1))


def keep_scale_reference(seed, idx, p):
    """numpy restatement of grad_core.h:keep_scale (murmur3 finaliser of seed ^ idx * golden ratio)"""
    import numpy as np
    m64 = (1 << 64) - 1
    x = (np.uint64(seed) ^ ((idx.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) & np.uint64(m64)))
    x ^= x >> np.uint64(33)
    x = (x * np.uint64(0xff51afd7ed558ccd)) & np.uint64(m64)
    x ^= x >> np.uint64(33)
    x = (x * np.uint64(0xc4ceb9fe1a85ec53)) & np.uint64(m64)
    x ^= x >> np.uint64(33)
    u = ((x & np.uint64(0xffffffff)) >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return np.where(u >= np.float32(p), np.float32(1.0) / (np.float32(1.0) - np.float32(p)),
                    np.float32(0.0)).astype(np.float32)


def test_dropout_mask(host):
    import numpy as np
    x = torch.randn(100000)
    out = torch.empty_like(x)
    seed, p = 123456789012345, 0.2
    assert host.host_dropout(P(x), P(out), x.numel(), p, seed, None) == 0
    with np.errstate(over="ignore"):
        ks = torch.from_numpy(keep_scale_reference(seed, np.arange(x.numel()), p))
    assert torch.equal(out, x * ks)
    kept = (out != 0).float().mean().item()
    assert abs(kept - (1 - p)) < 5e-3  # the keep rate
    again = torch.empty_like(x)
    assert host.host_dropout(P(x), P(again), x.numel(), p, seed, None) == 0
    assert torch.equal(again, out)  # the backward recomputes the same mask
    assert host.host_dropout(P(x), P(again), x.numel(), p, seed + 1, None) == 0
    assert not torch.equal(again, out)
    assert host.host_dropout(P(x), P(again), x.numel(), 0.0, seed, None) == 0
    assert torch.equal(again, x)


@pytest.mark.parametrize("use_rel,use_lens", [(False, False), (True, True)])
def test_attention_with_weight_dropout(host, use_rel, use_lens):
    """dropout on the attention weights (impl.py:104): training forward and the backward that
    recomputes the mask, against torch autograd with the same mask applied to softmax(S)"""
    import numpy as np
    torch.manual_seed(17)
    N, T, H, dh = 2, 9, 3, 32
    seed, p = 987654321, 0.3
    qkv = torch.randn(N, T, 3 * H * dh, requires_grad=True)
    lens = torch.tensor([9, 6]) if use_lens else None
    R = 2 * T - 1
    rel = torch.randn(R, dh, requires_grad=True) if use_rel else None
    zero = T - 1
    with np.errstate(over="ignore"):
        mask = torch.from_numpy(keep_scale_reference(seed, np.arange(N * H * T * T), p))
    mask = mask.view(N, H, T, T)
    # reference with the mask on the weights
    q, k, v = qkv.view(N, T, 3, H, dh).unbind(2)
    s = torch.einsum("nihd,njhd->nhij", q, k)
    if rel is not None:
        idx = torch.arange(T)[None, :] - torch.arange(T)[:, None] + zero
        s = s + torch.einsum("nihd,ijd->nhij", q, rel[idx])
    s = s / dh**0.5
    if lens is not None:
        s = s.masked_fill((torch.arange(T)[None, :] >= lens[:, None])[:, None, None, :],
                          float("-inf"))
    want = torch.einsum("nhij,njhd->nihd", torch.softmax(s, -1) * mask, v).reshape(N, T, -1)
    g = torch.randn_like(want)
    want.backward(g)
    ctx = torch.empty(N, T, H * dh)
    ws = torch.empty(host.host_attention_backward_workspace(N, T, H) // 4)
    rel_d = None if rel is None else rel.detach()
    rc = host.host_attention_forward_dropout(P(qkv.detach()), P(lens), P(rel_d), zero,
                                             R if use_rel else 0, P(ctx), N, T, H, dh, p, seed,
                                             P(ws), None)
    assert rc == 0
    close(ctx, want.detach(), what="attention forward with weight dropout")
    g_qkv = torch.empty(N, T, 3 * H * dh)
    part = torch.empty(N * H, R, dh) if use_rel else None
    rc = host.host_attention_backward(P(qkv.detach()), P(lens), P(rel_d), zero, R if use_rel else 0,
                                      P(g), P(g_qkv), P(part), N, T, H, dh, p, seed, P(ws), None)
    assert rc == 0
    close(g_qkv, qkv.grad, what="g_qkv (weight dropout)")
    if use_rel:
        close(part.sum(0), rel.grad, what="g_rel (weight dropout)")


def test_cplx_matmul_and_inverse_functors(host):
    """ComplexTensor.__matmul__ / inverse (aps/cplx.py:242-278): the functors behind aps_cplx_matmul /
    aps_cplx_inverse against numpy complex arithmetic, as the reference's own self-tests do
    (cplx.py:301-364: random 5 x 5 ... operands, matmul with complex and real right operands, inverse)"""
    import numpy as np
    rng = np.random.default_rng(5)
    B, M, K, N = 7, 5, 6, 4
    a = (rng.random((B, M, K)) + 1j * rng.random((B, M, K))).astype(np.complex64)
    for b, b_batch in (((rng.random((B, K, N)) + 1j * rng.random((B, K, N))).astype(np.complex64), K * N),
                       ((rng.random((K, N)) + 1j * rng.random((K, N))).astype(np.complex64), 0)):
        ar, ai = torch.from_numpy(a.real.copy()), torch.from_numpy(a.imag.copy())
        br, bi = torch.from_numpy(b.real.copy()), torch.from_numpy(b.imag.copy())
        cr, ci = torch.empty(B, M, N), torch.empty(B, M, N)
        assert host.host_cplx_matmul(P(ar), P(ai), P(br), P(bi), P(cr), P(ci), B, M, K, N, M * K, b_batch,
                                     None) == 0
        want = a @ b
        assert np.allclose(cr.numpy(), want.real, atol=1e-5) and np.allclose(ci.numpy(), want.imag, atol=1e-5)
        # a real right operand (imag half absent)
        assert host.host_cplx_matmul(P(ar), P(ai), P(br), None, P(cr), P(ci), B, M, K, N, M * K, b_batch,
                                     None) == 0
        want = a @ b.real
        assert np.allclose(cr.numpy(), want.real, atol=1e-5) and np.allclose(ci.numpy(), want.imag, atol=1e-5)
    for C in (1, 2, 4, 5, 8):
        m = (rng.random((B, C, C)) + 1j * rng.random((B, C, C)) + 2 * np.eye(C)).astype(np.complex64)
        mr, mi = torch.from_numpy(m.real.copy()), torch.from_numpy(m.imag.copy())
        orr, oi = torch.empty(B, C, C), torch.empty(B, C, C)
        count = torch.zeros(1, dtype=torch.int32)
        assert host.host_cplx_inverse(P(mr), P(mi), P(orr), P(oi), B, C, P(count), None) == 0
        want = np.linalg.inv(m.astype(np.complex128))
        got = orr.numpy() + 1j * oi.numpy()
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), C
        assert int(count) == 0
        # singular matrices are COUNTED (where th.inverse raises, aps/cplx.py:268-278): a zero matrix, one with a
        # zero column (C >= 2: an exactly zero pivot, like LAPACK's info > 0), one with a NaN -- and only those
        bad = m.copy()
        bad[0] = 0
        n_bad = 2
        if C >= 2:
            bad[2, :, 1] = 0
            n_bad = 3
        bad[4, 0, 0] = np.nan
        mr, mi = torch.from_numpy(bad.real.copy()), torch.from_numpy(bad.imag.copy())
        assert host.host_cplx_inverse(P(mr), P(mi), P(orr), P(oi), B, C, P(count), None) == 0
        assert int(count) == n_bad, (C, int(count))
        assert host.host_cplx_inverse(P(mr), P(mi), P(orr), P(oi), B, C, None, None) == 0   # (no counter: no check)
    assert host.host_cplx_inverse(P(mr), P(mi), P(orr), P(oi), B, 9, None, None) == -2   # APS_ERR_UNSUPPORTED


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("apply", [0, 1])
@pytest.mark.parametrize("nl", [0, 1, 2, 3, 4])
def test_dccrn_mask_backward(host, nl, apply, cplx):
    """aps_dccrn_mask_backward against autograd through the reference's masking arithmetic
    (aps/sse/bss/dccrn.py:217-242 as oracle/dccrn_oracle.py restates it)"""
    torch.manual_seed(nl * 4 + apply * 2 + cplx)
    S, rows, eps = 2, 300, 1.1920929e-07
    name = ["none", "relu", "tanh", "softplus", "sigmoid"][nl]
    fn = {"none": lambda v: v, "relu": torch.relu, "tanh": torch.tanh, "softplus": F.softplus,
          "sigmoid": torch.sigmoid}[name]
    dec = (torch.randn(rows, 2 * S if cplx else S) * 1.5).requires_grad_(True)
    store = torch.randn(rows, 2, requires_grad=True)
    sr, si = store[:, 0], store[:, 1]
    outs = []
    for s in range(S):
        if cplx:
            mr, mi = dec[:, s], dec[:, S + s]
            m_abs = (mr**2 + mi**2 + eps)**0.5
            m_mag = fn(m_abs)
            mr, mi = m_mag * mr / m_abs, m_mag * mi / m_abs
            outs.append(torch.stack([sr * mr - si * mi, sr * mi + si * mr], -1) if apply else
                        torch.stack([mr, mi], -1))
        else:
            m = fn(dec[:, s])
            outs.append(torch.stack([sr * m, si * m], -1) if apply else m)
    out = torch.stack(outs)
    g = torch.randn_like(out)
    out.backward(g)
    g_dec = torch.empty_like(dec)
    g_store = torch.empty(rows, 2) if apply else None
    rc = host.host_dccrn_mask_backward(P(dec.detach()), P(store.detach()) if apply else None, P(g),
                                       P(g_dec), P(g_store), rows, S, nl, apply, int(cplx), eps, None)
    assert rc == 0
    close(g_dec, dec.grad, tol=5e-5, what=f"mask backward g_dec ({name})")
    if apply:
        close(g_store, store.grad, tol=5e-5, what="mask backward g_store")
    # a gradient of the mixture's STFT without the masking is a caller error
    assert host.host_dccrn_mask_backward(P(dec.detach()), None, P(g), P(g_dec), P(torch.empty(rows, 2)),
                                         rows, S, nl, 0, int(cplx), eps, None) != 0


def test_colreduce_two_sums_in_one_call(host):
    """mode 4: the column sums of two matrices side by side (LayerNorm's g_gamma | g_beta)"""
    torch.manual_seed(31)
    for rows, D in [(2016, 96), (77, 5), (5000, 8)]:
        a, b = torch.randn(rows, D), torch.randn(rows, D)
        ws = torch.empty(host.host_colreduce_workspace(rows, 2 * D) // 4)
        out = torch.empty(2 * D)
        assert host.host_colreduce(4, P(a), P(b), None, None, rows, 2 * D, D, D, 1.0, 0, P(out), P(ws),
                                   None) == 0
        close(out[:D], a.double().sum(0).float(), what="first sums")
        close(out[D:], b.double().sum(0).float(), what="second sums")
    assert host.host_colreduce(4, P(a), P(b), None, None, rows, 2 * D + 1, D, D, 1.0, 0, P(out), P(ws),
                               None) != 0  # odd total: not two halves


def xl_window_reference(qkv, lens, table, u, v, zero, H, qslot, window, keep=None):
    """(q_src + u) . k + (q_src + v) . R_h[j - i + zero], context window, key padding -> context
    N x T x H dh (XlMultiheadAttention.dot_att, impl.py:322-374; prep_context_mask, utils.py:60-98);
    query rows without a visible key are left out by the caller"""
    N, T, D3 = qkv.shape
    dh = D3 // 3 // H
    parts = [m.reshape(N, T, H, dh) for m in qkv.chunk(3, -1)]
    src, key, val = parts[qslot], parts[1], parts[2]
    ac = torch.einsum("nlhd,nshd->nhls", src + (0 if u is None else u), key)
    score = ac
    if table is not None:
        R = table.shape[-2]
        idx = torch.arange(T)[None, :] - torch.arange(T)[:, None] + zero  # [i, j]
        ok = (idx >= 0) & (idx < R)
        tab = table if table.dim() == 3 else table[None].expand(H, -1, -1)
        gathered = tab[:, idx.clamp(0, R - 1)] * ok[None, :, :, None]  # H x T x T x dh
        score = score + torch.einsum("nlhd,hlsd->nhls", src + (0 if v is None else v), gathered)
    score = score / dh**0.5
    chunk, lctx, rctx = window
    i, j = torch.arange(T)[:, None], torch.arange(T)[None, :]
    cf = i // chunk
    vis = torch.ones(T, T, dtype=torch.bool)
    if rctx >= 0:
        vis &= j < (cf + rctx + 1) * chunk
    if lctx >= 0:
        vis &= j >= (cf - lctx) * chunk
    mask = ~vis[None, None]
    if lens is not None:
        mask = mask | (torch.arange(T)[None] >= lens[:, None])[:, None, None, :]
    score = score.masked_fill(mask, float("-inf"))
    prob = torch.softmax(score, -1)
    # a query whose window holds no valid key: NaN in the reference (a padded frame, never read), a
    # zero context row and no gradient here
    prob = torch.where(torch.isnan(prob), torch.zeros_like(prob), prob)
    if keep is not None:  # dropout on the weights: N x H x T x T factors 0 | 1 / (1 - p)
        prob = prob * keep
    return torch.einsum("nhls,nshd->nlhd", prob, val).reshape(N, T, H * dh)


@pytest.mark.parametrize("drop", [0.0, 0.25])
@pytest.mark.parametrize("case", ["window", "xl_shared", "xl_per_head_value_query", "xl_window"])
def test_attention_backward_xl(host, case, drop):
    """aps_attention_backward_xl (+ the training forward aps_attention_forward_xl_dropout): context
    windows, per-head tables, the XL biases, the query read from the value projection and dropout on
    the weights against autograd through the explicit float64 form"""
    import numpy as np
    torch.manual_seed(len(case))
    N, T, H, dh = 2, 10, 3, 8
    seed = 424242
    cfg = {"window": dict(window=(2, 1, 0)), "xl_shared": dict(xl=True, per_head=False),
           "xl_per_head_value_query": dict(xl=True, per_head=True, qslot=2),
           "xl_window": dict(xl=True, per_head=True, qslot=2, window=(1, 3, 1))}[case]
    window = cfg.get("window", (1, -1, -1))
    qslot = cfg.get("qslot", 0)
    keep = None
    if drop > 0:
        with np.errstate(over="ignore"):
            keep = torch.from_numpy(keep_scale_reference(seed, np.arange(N * H * T * T), drop))
        keep = keep.view(N, H, T, T).double()
    qkv = torch.randn(N, T, 3 * H * dh, dtype=torch.float64, requires_grad=True)
    lens = torch.tensor([T, 5])  # (the last queries of utterance 1 see no valid key through a window)
    R = 2 * T - 1 - 2
    zero = (R - 1) // 2
    table = u = v = None
    if cfg.get("xl") or case == "window":
        shape = (H, R, dh) if cfg.get("per_head") else (R, dh)
        table = torch.randn(*shape, dtype=torch.float64, requires_grad=True)
    if cfg.get("xl"):
        u = torch.randn(H, dh, dtype=torch.float64, requires_grad=True)
        v = torch.randn(H, dh, dtype=torch.float64, requires_grad=True)
    ctx = xl_window_reference(qkv, lens, table, u, v, zero, H, qslot, window, keep)
    g = torch.randn(N, T, H * dh, dtype=torch.float64)
    assert not torch.isnan(ctx).any()
    (ctx * g).sum().backward()
    f = lambda t: None if t is None else t.detach().float().contiguous()  # noqa: E731
    g_qkv = torch.empty(N, T, 3, H, dh)
    part = torch.empty(N * H, R, dh)
    row_k, row_e = torch.empty(N, T, H, dh), torch.empty(N, T, H, dh)
    ws = torch.empty(host.host_attention_backward_workspace(N, T, H) // 4)
    qf, tf, uf, vf, gf = f(qkv), f(table), f(u), f(v), f(g)
    stride = R * dh if cfg.get("per_head") else 0
    if drop > 0:
        out = torch.empty(N, T, H * dh)
        rc = host.host_attention_forward_xl_dropout(P(qf), P(lens), P(tf), zero, R, stride, P(uf), P(vf),
                                                    qslot, *window, P(out), N, T, H, dh, drop, seed, None, None)
        assert rc == 0
        close(out, ctx.detach().float(), what=f"{case} forward with weight dropout")
    rc = host.host_attention_backward_xl(P(qf), P(lens), P(tf), zero, R, stride,
                                         P(uf), P(vf), qslot, *window, P(gf), P(g_qkv), P(part), P(row_k),
                                         P(row_e), N, T, H, dh, drop, seed, None, P(ws), None)
    assert rc == 0
    if qslot == 2:  # the q slot carries the gradient of the scores' query row: it belongs to the v slot
        g_qkv[:, :, 2] += g_qkv[:, :, 0]
        g_qkv[:, :, 0] = 0
    close(g_qkv.reshape(N, T, -1), qkv.grad.float(), what=f"{case} g_qkv")
    want_tab = table.grad.float()
    got_tab = part.view(N, H, R, dh).sum(0) if cfg.get("per_head") else part.sum(0)
    close(got_tab, want_tab, what=f"{case} g_table")
    if u is not None:
        close(row_k.sum((0, 1)), u.grad.float(), what=f"{case} g_u")
        close(row_e.sum((0, 1)), v.grad.float(), what=f"{case} g_v")


@pytest.mark.parametrize("drop,use_lens", [(0.0, True), (0.3, True), (0.0, False)])
def test_cross_attention_forward_backward(host, drop, use_lens):
    """aps_attention_cross_forward_dropout / aps_attention_cross_backward (the decoder's attention over
    the encoder output, decoder.py:78-86) against autograd through softmax(q k^T / sqrt(dh)) (* mask) v"""
    import numpy as np
    torch.manual_seed(19)
    N, Tq, Tk, H, dh = 2, 7, 11, 3, 8
    seed = 99887766
    q = torch.randn(N, Tq, H * dh, dtype=torch.float64, requires_grad=True)
    kv = torch.randn(N, Tk, 2 * H * dh, dtype=torch.float64, requires_grad=True)
    lens = torch.tensor([Tk, 6]) if use_lens else None
    keep = torch.ones(N, H, Tq, Tk, dtype=torch.float64)
    if drop > 0:
        with np.errstate(over="ignore"):
            keep = torch.from_numpy(keep_scale_reference(seed, np.arange(N * H * Tq * Tk), drop))
        keep = keep.view(N, H, Tq, Tk).double()
    qh = q.view(N, Tq, H, dh)
    k, v = kv.view(N, Tk, 2, H, dh).unbind(2)
    s = torch.einsum("nihd,njhd->nhij", qh, k) / dh**0.5
    if lens is not None:
        s = s.masked_fill((torch.arange(Tk)[None] >= lens[:, None])[:, None, None, :], float("-inf"))
    want = torch.einsum("nhij,njhd->nihd", torch.softmax(s, -1) * keep, v).reshape(N, Tq, H * dh)
    g = torch.randn(N, Tq, H * dh, dtype=torch.float64)
    (want * g).sum().backward()
    qf, kvf, gf = (t.detach().float().contiguous() for t in (q, kv, g))
    ctx = torch.empty(N, Tq, H * dh)
    rc = host.host_attention_cross_forward_dropout(P(qf), P(kvf), P(lens), P(ctx), N, Tq, Tk, H, dh, drop,
                                                   seed, None, None)
    assert rc == 0
    close(ctx, want.detach().float(), what="cross attention forward")
    g_q, g_kv = torch.empty(N, Tq, H * dh), torch.empty(N, Tk, 2 * H * dh)
    ws = torch.empty(host.host_attention_cross_backward_workspace(N, Tq, H) // 4)
    rc = host.host_attention_cross_backward(P(qf), P(kvf), P(lens), P(gf), P(g_q), P(g_kv), N, Tq, Tk, H,
                                            dh, drop, seed, None, P(ws), None)
    assert rc == 0
    close(g_q, q.grad.float(), what="cross attention g_q")
    close(g_kv, kv.grad.float(), what="cross attention g_kv")


@pytest.mark.parametrize("drop", [0.0, 0.2])
@pytest.mark.parametrize("kind", ["causal", "bias", "causal+table"])
def test_attention_additive_mask_under_autograd(host, kind, drop):
    """an additive mask TENSOR on the scaled logits (the `src_mask` / `tgt_mask` a recipe passes in training:
    aps/asr/transformer/impl.py:104-114, decoder.py:150-186 -- a causal 0 / -inf mask, or any bias) through
    the general form: aps_attention_forward_xl_dropout / aps_attention_backward_xl with add_mask, against
    autograd through the explicit float64 form.  The mask is data: -inf pairs get weight and gradient 0, a
    query row that is masked everywhere (row 0 of utterance 1 has length 0 keys left after the mask and
    the lengths) gives a zero context row and no gradient."""
    import numpy as np
    torch.manual_seed(len(kind) + 3)
    N, T, H, dh = 2, 9, 2, 8
    seed = 13572468
    qkv = torch.randn(N, T, 3 * H * dh, dtype=torch.float64, requires_grad=True)
    lens = torch.tensor([T, 6])
    mask = torch.zeros(T, T, dtype=torch.float64)
    if "causal" in kind:
        mask = mask.masked_fill(torch.arange(T)[None, :] > torch.arange(T)[:, None], float("-inf"))
    if kind == "bias":
        mask = torch.randn(T, T, dtype=torch.float64)
        mask[3, :] = float("-inf")   # a query that sees nothing at all
    table, R, zero = None, 0, 0
    if "table" in kind:
        R = 2 * T - 1
        zero = T - 1
        table = torch.randn(R, dh, dtype=torch.float64, requires_grad=True)
    keep = None
    if drop > 0:
        with np.errstate(over="ignore"):
            keep = torch.from_numpy(keep_scale_reference(seed, np.arange(N * H * T * T), drop))
        keep = keep.view(N, H, T, T).double()
    # explicit form: scores as in xl_window_reference, + mask, key padding, softmax, (* keep), @ v
    parts = [m.reshape(N, T, H, dh) for m in qkv.chunk(3, -1)]
    score = torch.einsum("nlhd,nshd->nhls", parts[0], parts[1])
    if table is not None:
        idx = torch.arange(T)[None, :] - torch.arange(T)[:, None] + zero
        score = score + torch.einsum("nlhd,lsd->nhls", parts[0], table[idx])
    score = score / dh**0.5 + mask[None, None]
    score = score.masked_fill((torch.arange(T)[None] >= lens[:, None])[:, None, None, :], float("-inf"))
    dead = torch.isinf(score).all(-1, keepdim=True)   # (rows without a key: NaN in torch, zeros here)
    prob = torch.softmax(score.masked_fill(dead, 0.0), -1) * (~dead)
    if keep is not None:
        prob = prob * keep
    want = torch.einsum("nhls,nshd->nlhd", prob, parts[2]).reshape(N, T, H * dh)
    g = torch.randn(N, T, H * dh, dtype=torch.float64)
    (want * g).sum().backward()
    f = lambda t: None if t is None else t.detach().float().contiguous()  # noqa: E731
    qf, tf, gf, mf = f(qkv), f(table), f(g), f(mask)
    out = torch.empty(N, T, H * dh)
    rc = host.host_attention_forward_xl_dropout(P(qf), P(lens), P(tf), zero, R, 0, None, None, 0, 1, -1, -1,
                                                P(out), N, T, H, dh, drop, seed, P(mf), None)
    assert rc == 0
    close(out, want.detach().float(), what=f"{kind}: forward with the additive mask")
    g_qkv = torch.empty(N, T, 3, H, dh)
    part = torch.empty(N * H, max(R, 1), dh)
    ws = torch.empty(host.host_attention_backward_workspace(N, T, H) // 4)
    rc = host.host_attention_backward_xl(P(qf), P(lens), P(tf), zero, R, 0, None, None, 0, 1, -1, -1, P(gf),
                                         P(g_qkv), P(part) if table is not None else None, None, None, N, T, H, dh,
                                         drop, seed, P(mf), P(ws), None)
    assert rc == 0
    assert torch.isfinite(g_qkv).all()
    close(g_qkv.reshape(N, T, -1), qkv.grad.float(), what=f"{kind}: g_qkv")
    if table is not None:
        close(part.sum(0), table.grad.float(), what=f"{kind}: g_table")


@pytest.mark.parametrize("drop", [0.0, 0.3])
def test_cross_attention_memory_mask_under_autograd(host, drop):
    """the decoder layer's additive memory_mask [Tq, Tk] (decoder.py:150-186) in training: forward and
    backward of the cross attention with add_mask against float64 autograd; one query row masked everywhere"""
    import numpy as np
    torch.manual_seed(23)
    N, Tq, Tk, H, dh = 2, 5, 9, 2, 8
    seed = 24681357
    q = torch.randn(N, Tq, H * dh, dtype=torch.float64, requires_grad=True)
    kv = torch.randn(N, Tk, 2 * H * dh, dtype=torch.float64, requires_grad=True)
    lens = torch.tensor([Tk, 7])
    mask = torch.randn(Tq, Tk, dtype=torch.float64)
    mask[:, 0] = float("-inf")
    mask[2, :] = float("-inf")
    keep = torch.ones(N, H, Tq, Tk, dtype=torch.float64)
    if drop > 0:
        with np.errstate(over="ignore"):
            keep = torch.from_numpy(keep_scale_reference(seed, np.arange(N * H * Tq * Tk), drop))
        keep = keep.view(N, H, Tq, Tk).double()
    k, v = kv.view(N, Tk, 2, H, dh).unbind(2)
    s = torch.einsum("nihd,njhd->nhij", q.view(N, Tq, H, dh), k) / dh**0.5 + mask[None, None]
    s = s.masked_fill((torch.arange(Tk)[None] >= lens[:, None])[:, None, None, :], float("-inf"))
    dead = torch.isinf(s).all(-1, keepdim=True)
    prob = torch.softmax(s.masked_fill(dead, 0.0), -1) * (~dead)
    want = torch.einsum("nhij,njhd->nihd", prob * keep, v).reshape(N, Tq, H * dh)
    g = torch.randn(N, Tq, H * dh, dtype=torch.float64)
    (want * g).sum().backward()
    qf, kvf, gf, mf = (t.detach().float().contiguous() for t in (q, kv, g, mask))
    ctx = torch.empty(N, Tq, H * dh)
    assert host.host_attention_cross_forward_dropout(P(qf), P(kvf), P(lens), P(ctx), N, Tq, Tk, H, dh, drop, seed,
                                                     P(mf), None) == 0
    close(ctx, want.detach().float(), what="cross attention forward with memory_mask")
    assert float(ctx[:, 2].abs().max()) == 0
    g_q, g_kv = torch.empty(N, Tq, H * dh), torch.empty(N, Tk, 2 * H * dh)
    ws = torch.empty(host.host_attention_cross_backward_workspace(N, Tq, H) // 4)
    assert host.host_attention_cross_backward(P(qf), P(kvf), P(lens), P(gf), P(g_q), P(g_kv), N, Tq, Tk, H, dh, drop,
                                              seed, P(mf), P(ws), None) == 0
    assert torch.isfinite(g_q).all() and torch.isfinite(g_kv).all()
    close(g_q, q.grad.float(), what="cross attention g_q with memory_mask")
    close(g_kv, kv.grad.float(), what="cross attention g_kv with memory_mask")


def test_embedding_backward(host):
    """adjoint of the decoder's token embedding: rows summed per token over the sorted lookups"""
    torch.manual_seed(20)
    V, D, R = 13, 6, 40
    emb = torch.nn.Embedding(V, D)
    ids = torch.randint(0, V - 2, (R,))  # (the last two tokens never occur: zero rows)
    g = torch.randn(R, D)
    (emb(ids) * 0.5 * g).sum().backward()
    sorted_ids, order = torch.sort(ids, stable=True)
    gw = torch.zeros(V, D)
    rc = host.host_embedding_backward(P(sorted_ids), P(order), P(g), P(gw), R, D, V, 0.5, None)
    assert rc == 0
    close(gw, emb.weight.grad, what="embedding g_weight")
    assert float(gw[V - 2:].abs().max()) == 0


def test_cross_attention_without_valid_keys(host):
    """an utterance whose memory is all padding (key_lens = 0; NaN rows in torch, never read): zero
    context rows, no gradient to its query or memory, the other utterance untouched"""
    torch.manual_seed(21)
    N, Tq, Tk, H, dh = 2, 4, 6, 2, 8
    q, kv, g = torch.randn(N, Tq, H * dh), torch.randn(N, Tk, 2 * H * dh), torch.randn(N, Tq, H * dh)
    lens = torch.tensor([Tk, 0])
    ctx = torch.full((N, Tq, H * dh), 7.0)
    assert host.host_attention_cross_forward_dropout(P(q), P(kv), P(lens), P(ctx), N, Tq, Tk, H, dh, 0.0, 0,
                                                     None, None) == 0
    assert float(ctx[1].abs().max()) == 0 and float(ctx[0].abs().max()) > 0
    g_q, g_kv = torch.full_like(q, 7.0), torch.full_like(kv, 7.0)
    ws = torch.empty(host.host_attention_cross_backward_workspace(N, Tq, H) // 4)
    assert host.host_attention_cross_backward(P(q), P(kv), P(lens), P(g), P(g_q), P(g_kv), N, Tq, Tk, H, dh,
                                              0.0, 0, None, P(ws), None) == 0
    assert float(g_q[1].abs().max()) == 0 and float(g_kv[1].abs().max()) == 0
    assert float(g_q[0].abs().max()) > 0 and torch.isfinite(g_kv).all()


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("use_lens", [False, True])
def test_rnn_step_backward(host, mode, use_lens):
    """one step of nn.GRU / nn.RNN(tanh | relu) / nn.LSTM backwards (RnnStepBackward beside decoder.hip's
    rnn_step_kernel) against torch autograd through the cell formulas of torch.nn (the reference's
    var_len_rnn_forward runs those layers, component.py:26-55), packed-sequence rows included"""
    g = torch.Generator().manual_seed(20 + mode)
    N, H = 5, 7
    G = {0: 3, 1: 1, 2: 1, 3: 4}[mode]
    t = 2
    lens = torch.tensor([4, 2, 3, 1, 5]) if use_lens else None
    gx = torch.randn(N, G * H, generator=g).requires_grad_(True)
    gh = torch.randn(N, G * H, generator=g).requires_grad_(True)
    hp = torch.randn(N, H, generator=g).requires_grad_(True)
    cp = torch.randn(N, H, generator=g).requires_grad_(True)
    g_y, g_h, g_c = (torch.randn(N, H, generator=g) for _ in range(3))
    live = torch.ones(N, 1, dtype=torch.bool) if lens is None else (lens > t)[:, None]
    if mode == 0:
        r = torch.sigmoid(gx[:, :H] + gh[:, :H])
        z = torch.sigmoid(gx[:, H:2 * H] + gh[:, H:2 * H])
        n_ = torch.tanh(gx[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n_ + z * hp
        c = cp
    elif mode == 3:
        a = gx + gh
        i_, f_, g_, o_ = (a[:, k * H:(k + 1) * H] for k in range(4))
        c = torch.sigmoid(f_) * cp + torch.sigmoid(i_) * torch.tanh(g_)
        h = torch.sigmoid(o_) * torch.tanh(c)
    else:
        h = torch.tanh(gx + gh) if mode == 1 else torch.relu(gx + gh)
        c = cp
    h_state = torch.where(live, h, hp)
    c_state = torch.where(live, c, cp)
    y = torch.where(live, h, torch.zeros_like(h))
    loss = (y * g_y).sum() + (h_state * g_h).sum() + ((c_state * g_c).sum() if mode == 3 else 0)
    loss.backward()
    o_gx, o_gh = torch.full((N, G * H), 9.0), torch.full((N, G * H), 9.0)
    o_hp, o_cp = torch.empty(N, H), torch.empty(N, H)
    gxd, ghd, hpd, cpd = gx.detach(), gh.detach(), hp.detach(), cp.detach()
    rc = host.host_rnn_step_backward(P(gxd), G * H, P(ghd), P(hpd), P(cpd) if mode == 3 else None, P(lens), t,
                                     P(g_y), H, P(g_h), P(g_c) if mode == 3 else None, P(o_gx), P(o_gh), G * H,
                                     P(o_hp), P(o_cp) if mode == 3 else None, N, H, mode, None)
    assert rc == 0
    close(o_gx, gx.grad, what="g_gx")
    close(o_gh, gh.grad, what="g_gh")
    # the kernel's g_hp leaves out the recurrent term (the caller's GEMM adds g_gh W_hh): here gh is a leaf
    close(o_hp, hp.grad, what="g_h_prev (direct part)")
    if mode == 3:
        close(o_cp, cp.grad, what="g_c_prev")


@pytest.mark.parametrize("select", [False, True])
def test_fixed_beamformer_backward(host, select):
    """FixedBeamformer (aps/transform/enh.py:349-384) with trainable coefficients: input and coefficient
    gradients of the all-beams form and of one beam per utterance against autograd through the oracle"""
    g = torch.Generator().manual_seed(31)
    N, C, F, T, B = 3, 4, 6, 9, 5
    xr = torch.randn(N, C, F, T, generator=g).requires_grad_(True)
    xi = torch.randn(N, C, F, T, generator=g).requires_grad_(True)
    wr = torch.randn(B, C, F, generator=g).requires_grad_(True)
    wi = torch.randn(B, C, F, generator=g).requires_grad_(True)
    beam = torch.tensor([2, 0, 2]) if select else None
    br, bi = ao.fixed_beamform(xr, xi, wr, wi, beam)
    ur, ui = torch.randn(br.shape, generator=g), torch.randn(bi.shape, generator=g)
    ((br * ur).sum() + (bi * ui).sum()).backward()
    o = [torch.empty(N, C, F, T), torch.empty(N, C, F, T), torch.empty(B, C, F), torch.empty(B, C, F)]
    xrd, xid, wrd, wid = xr.detach(), xi.detach(), wr.detach(), wi.detach()
    rc = host.host_fixed_beamform_backward(P(ur), P(ui), P(xrd), P(xid), P(wrd), P(wid), P(beam), P(o[0]), P(o[1]),
                                           P(o[2]), P(o[3]), N, C, F, T, B, None)
    assert rc == 0
    close(o[0], xr.grad, what="g_x real")
    close(o[1], xi.grad, what="g_x imag")
    close(o[2], wr.grad, what="g_w real")
    close(o[3], wi.grad, what="g_w imag")
