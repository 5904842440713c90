"""
torch.autograd.Function wrappers over the backward entry points of include/aps_amd.h (grad.hip): what
makes the MVDR front end, its LSTM mask estimator, the transformer / conformer encoders (absolute,
relative and Transformer-XL positions, context windows, causal convolution, all three projections),
the transformer decoder, DCCRN and the feature chains (trainable mel filters, power spectrum)
trainable under the reference's trainer (`loss.backward()`, aps/trainer/ddp.py:161-165) -- section
8(f) row 1.

Every forward here is the HIP forward of the eval path (or its un-fused form where the backward needs
an intermediate the fused launch does not keep: the pre-activation of a GEMM, the un-normalised input
of a BatchNorm); every backward is HIP: index functors of grad_core.h plus launches of the forward's
fp32 MFMA GEMM on transposed operands.  There is no torch / CPU fallback: a case without a backward
kernel raises NotImplementedError.
"""
from typing import Optional

import torch as th

from aps_amd import _native as nat

ACT_CODES = {None: 0, "none": 0, "relu": 1, "swish": 2, "sigmoid": 3, "tanh": 4, "gelu": 5,
             "leaky_relu": 6, "square": 7}  # 6, 7: the stand-alone pass only (nn.LeakyReLU(), slope 0.01; x^2)


def _f32(t: th.Tensor) -> th.Tensor:
    return nat.f32c(t.detach())


def transpose2d(x: th.Tensor) -> th.Tensor:
    """[rows, cols] -> [cols, rows] contiguous (aps_transpose); x may have a row pitch"""
    rows, cols = x.shape
    if x.stride(1) != 1:
        x = x.contiguous()
    out = th.empty(cols, rows, device=x.device, dtype=th.float32)
    rc = nat.load().aps_transpose(nat.ptr(x), nat.ptr(out), rows, cols, x.stride(0), rows,
                                  nat.stream_of(x))
    nat.check(rc, "aps_transpose")
    return out


def colreduce(mode: int, A: th.Tensor, B: Optional[th.Tensor] = None,
              v1: Optional[th.Tensor] = None, v2: Optional[th.Tensor] = None, scale: float = 1.0,
              out: Optional[th.Tensor] = None) -> th.Tensor:
    """column reductions of [rows, cols] matrices (see aps_colreduce); out given = accumulate"""
    lib = nat.load()
    rows, cols = A.shape
    if A.stride(1) != 1:
        A = A.contiguous()
    if B is not None and B.stride(1) != 1:
        B = B.contiguous()
    ws = th.empty(lib.aps_colreduce_workspace(rows, cols) // 4, device=A.device, dtype=th.float32)
    acc = 0 if out is None else 1
    if out is None:
        out = th.empty(cols, device=A.device, dtype=th.float32)
    rc = lib.aps_colreduce(mode, nat.ptr(A), nat.ptr(B), nat.ptr(v1), nat.ptr(v2), rows, cols,
                           A.stride(0), 0 if B is None else B.stride(0), float(scale), acc,
                           nat.ptr(out), nat.ptr(ws), nat.stream_of(A))
    nat.check(rc, "aps_colreduce")
    return out


def colreduce2(A: th.Tensor, B: th.Tensor) -> th.Tensor:
    """column sums of two [rows, cols] matrices in one launch sequence -> [2 cols] (A's | B's)"""
    lib = nat.load()
    rows, cols = A.shape
    if tuple(B.shape) != (rows, cols):
        raise RuntimeError(f"colreduce2: {tuple(A.shape)} vs {tuple(B.shape)}")
    if A.stride(1) != 1:
        A = A.contiguous()
    if B.stride(1) != 1:
        B = B.contiguous()
    ws = th.empty(lib.aps_colreduce_workspace(rows, 2 * cols) // 4, device=A.device, dtype=th.float32)
    out = th.empty(2 * cols, device=A.device, dtype=th.float32)
    rc = lib.aps_colreduce(4, nat.ptr(A), nat.ptr(B), None, None, rows, 2 * cols, A.stride(0), B.stride(0),
                           1.0, 0, nat.ptr(out), nat.ptr(ws), nat.stream_of(A))
    nat.check(rc, "aps_colreduce")
    return out


def xty(x: th.Tensor, y: th.Tensor, colsum: bool = False):
    """x^T y of two tall row-major matrices [M, I], [M, J] -> [I, J] (aps_gemm_tn: fp32 MFMA straight
    from the operands as they lie, M cut into slabs summed in a fixed order); colsum: also the column
    sums of x -- the (weight, bias) gradient pair of a projection in one call"""
    lib = nat.load()
    M, I = x.shape
    J = y.shape[1]
    if y.shape[0] != M:
        raise RuntimeError(f"xty: {tuple(x.shape)} against {tuple(y.shape)}")
    if x.stride(1) != 1:
        x = x.contiguous()
    if y.stride(1) != 1:
        y = y.contiguous()
    most = (2 ** 31 - 1) // (4 * max(x.stride(0), y.stride(0)))  # rows a call's 32-bit byte offsets reach
    if M > most:  # (merged batches of the first conv2d layers: row blocks, summed)
        parts = [xty(x[r:r + most], y[r:r + most], colsum) for r in range(0, M, most)]
        if colsum:
            return sum(p[0] for p in parts), sum(p[1] for p in parts)
        return sum(parts)
    out = th.empty(I, J, device=x.device, dtype=th.float32)
    cs = th.empty(I, device=x.device, dtype=th.float32) if colsum else None
    nbytes = lib.aps_gemm_tn_workspace(M, I, J)
    ws = th.empty(nbytes // 4, device=x.device, dtype=th.float32) if nbytes else None
    rc = lib.aps_gemm_tn(nat.ptr(x), nat.ptr(y), nat.ptr(out), nat.ptr(cs), nat.ptr(ws), M, I, J,
                         x.stride(0), y.stride(0), J, nat.stream_of(x))
    nat.check(rc, "aps_gemm_tn")
    return (out, cs) if colsum else out


def _linear_nograd(x2d: th.Tensor, w: th.Tensor, bias: Optional[th.Tensor] = None) -> th.Tensor:
    from aps_amd import nn_ops
    with th.no_grad():
        return nn_ops.linear(x2d, w, bias)


def act_forward(pre: th.Tensor, residual: Optional[th.Tensor], act: int, alpha: float) -> th.Tensor:
    out = th.empty_like(pre)
    rc = nat.load().aps_act_forward(nat.ptr(pre), nat.ptr(residual), nat.ptr(out), pre.numel(), act,
                                    float(alpha), nat.stream_of(pre))
    nat.check(rc, "aps_act_forward")
    return out


def act_backward(g: th.Tensor, pre: th.Tensor, act: int, alpha: float) -> th.Tensor:
    out = th.empty_like(pre)
    rc = nat.load().aps_act_backward(nat.ptr(g), nat.ptr(pre), nat.ptr(out), pre.numel(), act,
                                     float(alpha), nat.stream_of(pre))
    nat.check(rc, "aps_act_backward")
    return out


class LinearFn(th.autograd.Function):
    """y = act(x W^T + b) * alpha (+ residual); backward: g_x = g_pre W (the forward GEMM on W^T),
    g_W = g_pre^T x and g_b = column sums of g_pre in one aps_gemm_tn call"""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, act, alpha):
        K, N = x.shape[-1], weight.shape[0]
        x2 = _f32(x).reshape(-1, K)
        w = _f32(weight)
        pre = _linear_nograd(x2, w, None if bias is None else _f32(bias))
        plain = act == 0 and alpha == 1.0
        if plain and residual is None:
            out = pre
        else:
            res = None if residual is None else _f32(residual).reshape(-1, N)
            out = act_forward(pre, res, act, alpha)
        ctx.save_for_backward(x2, w, None if plain else pre)
        ctx.cfg = (act, alpha, tuple(x.shape), bias is not None, residual is not None)
        return out.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        x2, w, pre = ctx.saved_tensors
        act, alpha, xshape, has_bias, has_res = ctx.cfg
        N = w.shape[0]
        g2 = nat.f32c(g).reshape(-1, N)
        g_res = g if has_res and ctx.needs_input_grad[3] else None
        g_pre = g2 if pre is None else act_backward(g2, pre, act, alpha)
        g_x = g_w = g_b = None
        if ctx.needs_input_grad[0]:
            g_x = _linear_nograd(g_pre, transpose2d(w)).view(xshape)
        want_b = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            g_w = xty(g_pre, x2, colsum=want_b)
            if want_b:
                g_w, g_b = g_w
        elif want_b:
            g_b = colreduce(0, g_pre)
        return g_x, g_w, g_b, g_res, None, None


class ActivationFn(th.autograd.Function):
    """stand-alone activation (behind a training-mode BatchNorm)"""

    @staticmethod
    def forward(ctx, x, act):
        xc = _f32(x)
        ctx.save_for_backward(xc)
        ctx.act = act
        return act_forward(xc, None, act, 1.0)

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        return act_backward(nat.f32c(g), xc, ctx.act, 1.0), None


def activation(x: th.Tensor, act: Optional[str]) -> th.Tensor:
    code = ACT_CODES[act]
    return x if code == 0 else ActivationFn.apply(x, code)


def draw_seed() -> int:
    """a dropout seed from torch's CPU generator (torch.manual_seed makes runs reproducible; no GPU
    synchronisation)"""
    return int(th.randint(0, 2**62, (1,)).item())


class DropoutFn(th.autograd.Function):
    """nn.Dropout in train() mode: counter-based mask, recomputed in the backward (aps_dropout)"""

    @staticmethod
    def forward(ctx, x, p, seed):
        xc = _f32(x)
        out = th.empty_like(xc)
        nat.check(nat.load().aps_dropout(nat.ptr(xc), nat.ptr(out), xc.numel(), float(p), int(seed),
                                         nat.stream_of(xc)), "aps_dropout")
        ctx.cfg = (float(p), int(seed))
        return out

    @staticmethod
    def backward(ctx, g):
        p, seed = ctx.cfg
        g = nat.f32c(g)
        out = th.empty_like(g)
        nat.check(nat.load().aps_dropout(nat.ptr(g), nat.ptr(out), g.numel(), p, seed,
                                         nat.stream_of(g)), "aps_dropout")
        return out, None, None


def dropout(x: th.Tensor, module: th.nn.Dropout) -> th.Tensor:
    """`module(x)` for an nn.Dropout: identity in eval mode / p = 0"""
    if not module.training or module.p <= 0 or x.numel() == 0:
        return x
    if module.p >= 1:
        raise ValueError("dropout probability must be < 1")
    return DropoutFn.apply(x, module.p, draw_seed())


def dropout_active(*modules) -> bool:
    return any(m is not None and m.training and m.p > 0 for m in modules)


class ScaleAddFn(th.autograd.Function):
    """x * alpha + residual (the residual connection behind a dropout, where the GEMM epilogue
    cannot carry it)"""

    @staticmethod
    def forward(ctx, x, residual, alpha):
        ctx.alpha = float(alpha)
        return act_forward(_f32(x), _f32(residual), 0, float(alpha))

    @staticmethod
    def backward(ctx, g):
        g = nat.f32c(g)
        ga = g if ctx.alpha == 1.0 else act_forward(g, None, 0, ctx.alpha)
        return ga, g, None


class RowBiasAddFn(th.autograd.Function):
    """x (..., D) + b [D]"""

    @staticmethod
    def forward(ctx, x, b):
        xc, bc = _f32(x), _f32(b)
        D = xc.shape[-1]
        out = th.empty_like(xc)
        nat.check(nat.load().aps_row_bias_add(nat.ptr(xc), nat.ptr(bc), nat.ptr(out),
                                              xc.numel() // D, D, nat.stream_of(xc)),
                  "aps_row_bias_add")
        return out

    @staticmethod
    def backward(ctx, g):
        g = nat.f32c(g)
        D = g.shape[-1]
        return g, colreduce(0, g.view(-1, D))


class GatherRowsFn(th.autograd.Function):
    """weight[index] (rows of an nn.Embedding table) with the scatter-add adjoint"""

    @staticmethod
    def forward(ctx, weight, index):
        ctx.save_for_backward(index)
        ctx.V = weight.shape[0]
        return weight.detach()[index]  # a row gather of a <= 2T-1 row table: torch indexing

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        g = nat.f32c(g)
        R, D = g.shape
        g_w = th.empty(ctx.V, D, device=g.device, dtype=th.float32)
        idx = index.to(th.int64).contiguous()
        nat.check(nat.load().aps_gather_rows_backward(nat.ptr(idx), nat.ptr(g), nat.ptr(g_w), R,
                                                      ctx.V, D, nat.stream_of(g)),
                  "aps_gather_rows_backward")
        return g_w, None


class LayerNormFn(th.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, eps):
        from aps_amd import nn_ops
        with th.no_grad():
            out = nn_ops.layernorm(x.detach(), weight.detach(), bias.detach(), eps,
                                   residual=None if residual is None else residual.detach())
        ctx.save_for_backward(_f32(x), None if residual is None else _f32(residual), _f32(weight))
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, g):
        x, res, gamma = ctx.saved_tensors
        D = x.shape[-1]
        rows = x.numel() // D
        g = nat.f32c(g)
        g_x = th.empty_like(x)
        t = th.empty(rows, D, device=x.device, dtype=th.float32)
        rc = nat.load().aps_layernorm_backward(nat.ptr(x), nat.ptr(res), nat.ptr(gamma), nat.ptr(g),
                                               nat.ptr(g_x), nat.ptr(t), rows, D, float(ctx.eps),
                                               nat.stream_of(x))
        nat.check(rc, "aps_layernorm_backward")
        g_gamma = g_beta = None
        if ctx.needs_input_grad[2] and ctx.needs_input_grad[3]:
            both = colreduce2(t, g.view(rows, D))  # g_gamma | g_beta in one pair of launches
            g_gamma, g_beta = both[:D], both[D:]
        elif ctx.needs_input_grad[2]:
            g_gamma = colreduce(0, t)
        elif ctx.needs_input_grad[3]:
            g_beta = colreduce(0, g.view(rows, D))
        return g_x, (g_x if res is not None else None), g_gamma, g_beta, None


class BatchNormRowsFn(th.autograd.Function):
    """BatchNorm over the rows of [..., D] (channels last): batch statistics + running-statistics
    update in training mode, running statistics as constants in eval mode"""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps):
        lib = nat.load()
        xc = _f32(x)
        D = xc.shape[-1]
        rows = xc.numel() // D
        st = nat.stream_of(xc)
        if training:
            mean = th.empty(D, device=xc.device, dtype=th.float32)
            rstd = th.empty(D, device=xc.device, dtype=th.float32)
            ws = th.empty(lib.aps_batchnorm_workspace(rows, D) // 4, device=xc.device,
                          dtype=th.float32)
            rc = lib.aps_batchnorm_stats(nat.ptr(xc), rows, D, float(eps), float(momentum),
                                         nat.ptr(mean), nat.ptr(rstd), nat.ptr(running_mean),
                                         nat.ptr(running_var), nat.ptr(ws), st)
            nat.check(rc, "aps_batchnorm_stats")
        else:
            mean = running_mean.detach().float()
            rstd = th.rsqrt(running_var.detach().float() + eps)  # [D] vector: plumbing
        y = th.empty_like(xc)
        gam = None if weight is None else _f32(weight)
        rc = lib.aps_batchnorm_apply(nat.ptr(xc), nat.ptr(mean), nat.ptr(rstd), nat.ptr(gam),
                                     nat.ptr(None if bias is None else _f32(bias)), nat.ptr(y), rows,
                                     D, st)
        nat.check(rc, "aps_batchnorm_apply")
        ctx.save_for_backward(xc, mean, rstd, gam)
        ctx.training = training
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        xc, mean, rstd, gam = ctx.saved_tensors
        D = xc.shape[-1]
        rows = xc.numel() // D
        g = nat.f32c(g)
        g2, x2 = g.view(rows, D), xc.view(rows, D)
        s1 = colreduce(0, g2)
        s2 = colreduce(3, g2, x2, mean, rstd)
        g_x = th.empty_like(xc)
        rc = nat.load().aps_batchnorm_backward(nat.ptr(xc), nat.ptr(mean), nat.ptr(rstd),
                                               nat.ptr(gam), nat.ptr(g),
                                               nat.ptr(s1 if ctx.training else None),
                                               nat.ptr(s2 if ctx.training else None), nat.ptr(g_x),
                                               rows, D, nat.stream_of(xc))
        nat.check(rc, "aps_batchnorm_backward")
        # (autograd rejects a gradient for an input that was None: affine=False, or a bias-free affine)
        return g_x, (s2 if gam is not None else None), (s1 if ctx.has_bias else None), None, None, None, \
            None, None


def batchnorm_rows(x: th.Tensor, bn: th.nn.modules.batchnorm._BatchNorm) -> th.Tensor:
    """nn.BatchNorm1d / 2d on channels-last activations (..., D)"""
    training = bn.training or bn.running_mean is None
    if training and bn.running_mean is not None and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    if bn.momentum is not None:
        momentum = bn.momentum
    elif training and bn.running_mean is not None and bn.num_batches_tracked is not None:
        # momentum=None: torch's cumulative moving average, factor 1 / num_batches_tracked
        # (nn/modules/batchnorm.py; the count was just advanced).  A host read once per call, like
        # torch's own `float(self.num_batches_tracked)`
        momentum = 1.0 / float(bn.num_batches_tracked)
    else:
        momentum = 0.0  # (no running statistics to update)
    return BatchNormRowsFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training,
                                 momentum, bn.eps)


def row_affine(x: th.Tensor, weight: Optional[th.Tensor], bias: Optional[th.Tensor]) -> th.Tensor:
    """x [..., D] * weight [D] + bias [D]: the BatchNorm apply / backward kernels with mean 0, rstd 1"""
    if weight is None and bias is None:
        return x
    D = x.shape[-1]
    return BatchNormRowsFn.apply(x, weight, bias, th.zeros(D, device=x.device), th.ones(D, device=x.device),
                                 False, 0.0, 0.0)


class UtteranceNormFn(th.autograd.Function):
    """(x - mean) / sqrt(var + eps) with the statistics of each utterance's WHOLE T x D matrix:
    GroupNorm(1, D) on N x D x T without its affine (Normalize1d "LN", component.py:85-114).  Forward
    = aps_cmvn_utterance; backward = the LayerNorm adjoint on rows of T D values."""

    @staticmethod
    def forward(ctx, x, eps):
        from aps_amd import ops
        xc = _f32(x)
        with th.no_grad():
            out = ops.cmvn_utterance(xc, True, True, eps)
        ctx.save_for_backward(xc)
        ctx.eps = float(eps)
        return out

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        count = xc.shape[-1] * xc.shape[-2]
        g_x = th.empty_like(xc)
        rc = nat.load().aps_layernorm_backward(nat.ptr(xc), None, None, nat.ptr(nat.f32c(g)),
                                               nat.ptr(g_x), None, xc.numel() // count, count, ctx.eps,
                                               nat.stream_of(xc))
        nat.check(rc, "aps_layernorm_backward")
        return g_x, None


class AttentionFn(th.autograd.Function):
    """aps_attention_core with absolute / learnt relative positions and length masks; drop_p > 0:
    dropout on the attention weights (training forward aps_attention_forward_dropout)"""

    @staticmethod
    def forward(ctx, qkv, rel, lens, num_heads, rel_zero, drop_p, drop_seed):
        from aps_amd import nn_ops
        qc = _f32(qkv)
        rc = None if rel is None else _f32(rel)
        if rc is not None and rel_zero is None:
            rel_zero = (rc.shape[-2] - 1) // 2
        if lens is not None:
            lens = lens.to(device=qc.device, dtype=th.int64).contiguous()
        if drop_p > 0:
            lib = nat.load()
            N, T, D3 = qc.shape
            dh = D3 // 3 // num_heads
            if rc is not None and rc.dim() != 2:
                raise NotImplementedError("aps_amd: attention dropout with per-head relative tables")
            out = th.empty(N, T, D3 // 3, device=qc.device, dtype=th.float32)
            ws = th.empty(lib.aps_attention_backward_workspace(N, T, num_heads) // 4,
                          device=qc.device, dtype=th.float32)
            rc_ = lib.aps_attention_forward_dropout(nat.ptr(qc), nat.ptr(lens), nat.ptr(rc),
                                                    int(rel_zero or 0),
                                                    0 if rc is None else rc.shape[0], nat.ptr(out), N,
                                                    T, num_heads, dh, float(drop_p), int(drop_seed),
                                                    nat.ptr(ws), nat.stream_of(qc))
            nat.check(rc_, "aps_attention_forward_dropout")
        else:
            with th.no_grad():
                out = nn_ops.attention_core(qc, num_heads, lens, rel=rc, rel_zero=rel_zero)
        ctx.save_for_backward(qc, rc, lens)
        ctx.cfg = (num_heads, rel_zero, float(drop_p), int(drop_seed))
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, rel, lens = ctx.saved_tensors
        H, rel_zero, drop_p, drop_seed = ctx.cfg
        lib = nat.load()
        N, T, D3 = qkv.shape
        dh = D3 // 3 // H
        g = nat.f32c(g)
        g_qkv = th.empty_like(qkv)
        ws = th.empty(lib.aps_attention_backward_workspace(N, T, H) // 4, device=qkv.device,
                      dtype=th.float32)
        R = 0 if rel is None else rel.shape[0]
        if rel is not None and rel.dim() != 2:
            raise NotImplementedError("aps_amd: attention backward with per-head relative tables")
        part = None if rel is None else th.empty(N * H, R * dh, device=qkv.device, dtype=th.float32)
        rc = lib.aps_attention_backward(nat.ptr(qkv), nat.ptr(lens), nat.ptr(rel),
                                        int(rel_zero or 0), R, nat.ptr(g), nat.ptr(g_qkv),
                                        nat.ptr(part), N, T, H, dh, drop_p, drop_seed, nat.ptr(ws),
                                        nat.stream_of(qkv))
        nat.check(rc, "aps_attention_backward")
        g_rel = None
        if rel is not None and ctx.needs_input_grad[1]:
            g_rel = colreduce(0, part).view(R, dh)
        return g_qkv, g_rel, None, None, None, None, None


class AttentionXlFn(th.autograd.Function):
    """aps_attention_core in its general form -- context windows (chunk_size / lctx / rctx), per-head
    relative tables, the Transformer-XL biases, the query read from the value projection -- with the
    adjoint aps_attention_backward_xl (generic kernels); drop_p > 0: dropout on the attention weights
    (training forward aps_attention_forward_xl_dropout, the backward recomputes the mask)"""

    @staticmethod
    def forward(ctx, qkv, rel, rel_u, rel_v, lens, num_heads, rel_zero, query_from_value, chunk, lctx,
                rctx, drop_p=0.0, drop_seed=0, add_mask=None):
        from aps_amd import nn_ops
        qc = _f32(qkv)
        mk = None if add_mask is None else _f32(add_mask.detach())  # data: no gradient into a mask
        rc = None if rel is None else _f32(rel)
        uc = None if rel_u is None else _f32(rel_u)
        vc = None if rel_v is None else _f32(rel_v)
        if rc is not None and rel_zero is None:
            rel_zero = (rc.shape[-2] - 1) // 2
        if lens is not None:
            lens = lens.to(device=qc.device, dtype=th.int64).contiguous()
        if drop_p > 0:
            N, T, D3 = qc.shape
            dh = D3 // 3 // num_heads
            R = 0 if rc is None else rc.shape[-2]
            out = th.empty(N, T, D3 // 3, device=qc.device, dtype=th.float32)
            rc_ = nat.load().aps_attention_forward_xl_dropout(
                nat.ptr(qc), nat.ptr(lens), nat.ptr(rc), int(rel_zero or 0), R,
                R * dh if rc is not None and rc.dim() == 3 else 0, nat.ptr(uc), nat.ptr(vc),
                2 if query_from_value else 0, int(chunk), int(lctx), int(rctx), nat.ptr(out), N, T,
                num_heads, dh, float(drop_p), int(drop_seed), nat.ptr(mk), nat.stream_of(qc))
            nat.check(rc_, "aps_attention_forward_xl_dropout")
        else:
            with th.no_grad():
                out = nn_ops.attention_core(qc, num_heads, lens, rel=rc, rel_zero=rel_zero, rel_u=uc,
                                            rel_v=vc, query_from_value=query_from_value,
                                            chunk_size=chunk, lctx=lctx, rctx=rctx, add_mask=mk)
        ctx.save_for_backward(qc, rc, uc, vc, lens, mk)
        ctx.cfg = (num_heads, rel_zero, bool(query_from_value), int(chunk), int(lctx), int(rctx),
                   float(drop_p), int(drop_seed))
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, rel, u, v, lens, mk = ctx.saved_tensors
        H, rel_zero, from_value, chunk, lctx, rctx, drop_p, drop_seed = ctx.cfg
        lib = nat.load()
        N, T, D3 = qkv.shape
        dh = D3 // 3 // H
        g = nat.f32c(g)
        g_qkv = th.empty(N, T, 3, H, dh, device=qkv.device, dtype=th.float32)
        ws = th.empty(lib.aps_attention_backward_workspace(N, T, H) // 4, device=qkv.device,
                      dtype=th.float32)
        R = 0 if rel is None else rel.shape[-2]
        per_head = rel is not None and rel.dim() == 3
        part = None if rel is None else th.empty(N * H, R * dh, device=qkv.device, dtype=th.float32)
        xl = u is not None or v is not None
        row_k = th.empty(N * T, H * dh, device=qkv.device, dtype=th.float32) if xl else None
        row_e = th.empty(N * T, H * dh, device=qkv.device, dtype=th.float32) if xl else None
        rc = lib.aps_attention_backward_xl(nat.ptr(qkv), nat.ptr(lens), nat.ptr(rel), int(rel_zero or 0),
                                           R, R * dh if per_head else 0, nat.ptr(u), nat.ptr(v),
                                           2 if from_value else 0, chunk, lctx, rctx, nat.ptr(g),
                                           nat.ptr(g_qkv), nat.ptr(part), nat.ptr(row_k), nat.ptr(row_e),
                                           N, T, H, dh, drop_p, drop_seed, nat.ptr(mk), nat.ptr(ws),
                                           nat.stream_of(qkv))
        nat.check(rc, "aps_attention_backward_xl")
        if from_value:  # the scores' query row was the value projection: its gradient belongs there
            g_qkv[:, :, 2] += g_qkv[:, :, 0]
            g_qkv[:, :, 0] = 0
        g_rel = g_u = g_v = None
        if rel is not None and ctx.needs_input_grad[1]:
            if per_head:  # sum over the utterances only
                g_rel = colreduce(0, part.view(N, H * R * dh)).view(H, R, dh)
            else:
                g_rel = colreduce(0, part).view(R, dh)
        if u is not None and ctx.needs_input_grad[2]:
            g_u = colreduce(0, row_k).view(H, dh)
        if v is not None and ctx.needs_input_grad[3]:
            g_v = colreduce(0, row_e).view(H, dh)
        return g_qkv.view(N, T, D3), g_rel, g_u, g_v, None, None, None, None, None, None, None, None, None, None


class AttentionCrossFn(th.autograd.Function):
    """aps_attention_cross (the decoder's attention over the encoder output) with length masks;
    drop_p > 0: dropout on the attention weights (training forward
    aps_attention_cross_forward_dropout, the backward recomputes the mask)"""

    @staticmethod
    def forward(ctx, q, kv, key_lens, num_heads, drop_p, drop_seed, add_mask=None):
        from aps_amd import nn_ops
        qc, kc = _f32(q), _f32(kv)
        mk = None if add_mask is None else _f32(add_mask.detach())  # data: no gradient into a mask
        if key_lens is not None:
            key_lens = key_lens.to(device=qc.device, dtype=th.int64).contiguous()
        N, Tq, D = qc.shape
        Tk = kc.shape[1]
        if drop_p > 0:
            out = th.empty(N, Tq, D, device=qc.device, dtype=th.float32)
            rc = nat.load().aps_attention_cross_forward_dropout(
                nat.ptr(qc), nat.ptr(kc), nat.ptr(key_lens), nat.ptr(out), N, Tq, Tk, num_heads,
                D // num_heads, float(drop_p), int(drop_seed), nat.ptr(mk), nat.stream_of(qc))
            nat.check(rc, "aps_attention_cross_forward_dropout")
        else:
            with th.no_grad():
                out = nn_ops.attention_cross(qc, kc, num_heads, key_lens, add_mask=mk)
        ctx.save_for_backward(qc, kc, key_lens, mk)
        ctx.cfg = (num_heads, float(drop_p), int(drop_seed))
        return out

    @staticmethod
    def backward(ctx, g):
        q, kv, key_lens, mk = ctx.saved_tensors
        H, drop_p, drop_seed = ctx.cfg
        lib = nat.load()
        N, Tq, D = q.shape
        Tk = kv.shape[1]
        g_q, g_kv = th.empty_like(q), th.empty_like(kv)
        ws = th.empty(lib.aps_attention_cross_backward_workspace(N, Tq, H) // 4, device=q.device,
                      dtype=th.float32)
        rc = lib.aps_attention_cross_backward(nat.ptr(q), nat.ptr(kv), nat.ptr(key_lens),
                                              nat.ptr(nat.f32c(g)), nat.ptr(g_q), nat.ptr(g_kv), N, Tq,
                                              Tk, H, D // H, drop_p, drop_seed, nat.ptr(mk), nat.ptr(ws),
                                              nat.stream_of(q))
        nat.check(rc, "aps_attention_cross_backward")
        return g_q, g_kv, None, None, None, None, None


class EmbeddingPosencFn(th.autograd.Function):
    """table[ids] * factor + sinusoid (aps_embedding_posenc: the decoder's token embedding) with the
    adjoint w.r.t. the table: the lookups sorted by token (index plumbing: torch.sort), the rows of
    each token summed in that order by aps_embedding_backward"""

    @staticmethod
    def forward(ctx, table, ids, div_term, factor, t0):
        from aps_amd import nn_ops
        with th.no_grad():
            out = nn_ops.embedding_posenc(table.detach(), ids, div_term.detach(), factor, t0)
        ctx.save_for_backward(ids)
        ctx.cfg = (tuple(table.shape), float(factor))
        return out

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        (V, D), factor = ctx.cfg
        g = nat.f32c(g).reshape(-1, D)
        sorted_ids, order = th.sort(ids.reshape(-1).to(th.int64), stable=True)
        g_w = th.zeros(V, D, device=g.device, dtype=th.float32)
        rc = nat.load().aps_embedding_backward(nat.ptr(sorted_ids.contiguous()), nat.ptr(order.contiguous()),
                                               nat.ptr(g), nat.ptr(g_w), g.shape[0], D, V, factor,
                                               nat.stream_of(g))
        nat.check(rc, "aps_embedding_backward")
        return g_w, None, None, None, None


class GluDwconvFn(th.autograd.Function):
    """GLU -> depthwise Conv1d (+ bias): aps_glu_dwconv without the BatchNorm affine / activation;
    causal: K - 1 frames of left context that carry glu(pad_bias) (zeros without pad_bias)"""

    @staticmethod
    def forward(ctx, x, weight, bias, causal=False, pad_bias=None):
        from aps_amd import nn_ops
        with th.no_grad():
            out = nn_ops.glu_dwconv(x.detach(), weight.detach(),
                                    None if bias is None else bias.detach(), None, None, act="none",
                                    causal=causal,
                                    pad_bias=None if pad_bias is None else pad_bias.detach())
        D = x.shape[-1] // 2
        ctx.save_for_backward(_f32(x), _f32(weight).reshape(D, -1),
                              None if pad_bias is None else _f32(pad_bias))
        ctx.has_bias = bias is not None
        ctx.causal = bool(causal)
        ctx.wshape = tuple(weight.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, pad_bias = ctx.saved_tensors
        lib = nat.load()
        N, T, D2 = x.shape
        D, K = D2 // 2, w.shape[1]
        g = nat.f32c(g)
        g_x = th.empty_like(x)
        g_w = th.empty(D, K, device=x.device, dtype=th.float32)
        ws = th.empty(lib.aps_glu_dwconv_backward_workspace(N, T, D, K) // 4, device=x.device,
                      dtype=th.float32)
        g_pad = None
        if ctx.causal:
            if pad_bias is not None:
                g_pad = th.empty(2 * D, device=x.device, dtype=th.float32)
            rc = lib.aps_glu_dwconv_backward_causal(
                nat.ptr(x), nat.ptr(w), nat.ptr(g), nat.ptr(pad_bias), nat.ptr(g_x), nat.ptr(g_w),
                nat.ptr(g_pad), N, T, D, K, nat.ptr(ws), nat.stream_of(x))
            nat.check(rc, "aps_glu_dwconv_backward_causal")
        else:
            rc = lib.aps_glu_dwconv_backward(nat.ptr(x), nat.ptr(w), nat.ptr(g), nat.ptr(g_x),
                                             nat.ptr(g_w), N, T, D, K, nat.ptr(ws), nat.stream_of(x))
            nat.check(rc, "aps_glu_dwconv_backward")
        g_b = colreduce(0, g.view(N * T, D)) if ctx.has_bias else None
        return g_x, g_w.view(ctx.wshape), g_b, None, g_pad


def _conv_weight_grad(inp: th.Tensor, g_out: th.Tensor, KH: int, KW: int, stride, padding):
    """weight gradient of the forward convolution inp [N, H, W, Ci] -> g_out's shape [N, Ho, Wo, Co]:
    g_out^T im2col(inp) (one GEMM) -> Co x KH x KW x Ci"""
    lib = nat.load()
    N, H, W, Ci = inp.shape
    _, Ho, Wo, Co = g_out.shape
    (sh, sw), (ph, pw) = stride, padding
    kk = KH * KW * Ci
    ld = (kk + 3) // 4 * 4
    M = N * Ho * Wo
    patches = th.empty(M, ld, device=inp.device, dtype=th.float32)
    rc = lib.aps_im2col_nhwc(nat.ptr(inp), nat.ptr(patches), N, H, W, Ci, KH, KW, sh, sw, ph, pw, Ho,
                             Wo, ld, nat.stream_of(inp))
    nat.check(rc, "aps_im2col_nhwc")
    g_w = xty(g_out.view(M, Co), patches)  # [Co, ld]
    return g_w[:, :kk].reshape(Co, KH, KW, Ci)


class Conv2dNhwcFn(th.autograd.Function):
    """channels-last Conv2d / ConvTranspose2d + bias (no BatchNorm scale, no activation): x
    N x H x W x Ci, w Co x KH x KW x Ci (the kernel's layout for either form), `crop` trailing output
    rows / columns not computed (the causal blocks' truncation).

    forward form F:    g_x = the transposed form of the same kernel on g_y,  g_w = g_y^T im2col(x)
    transposed form:   y = F'^T x with F' the forward convolution of weight w' = w.permute(3,1,2,0)
                       (Ho x Wo x Co maps -> H x W x Ci maps), so  g_x = F'(g_y)  and  g_w' is F''s
                       weight gradient with the roles of input and output gradient taken by g_y, x
    """

    @staticmethod
    def forward(ctx, x, w, bias, stride, padding, transposed, output_padding, crop):
        from aps_amd import nn_ops
        with th.no_grad():
            out = nn_ops.conv2d_nhwc(x.detach(), w.detach(), None,
                                     None if bias is None else bias.detach(), stride=stride,
                                     padding=padding, transposed=transposed,
                                     output_padding=output_padding, crop=crop)
        ctx.save_for_backward(_f32(x), _f32(w))
        ctx.cfg = (tuple(stride), tuple(padding), bool(transposed), tuple(output_padding),
                   tuple(crop), bias is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        from aps_amd import nn_ops
        x, w = ctx.saved_tensors
        (sh, sw), (ph, pw), transposed, _, (ch, cw), has_bias = ctx.cfg
        N, H, W, Ci = x.shape
        Co, KH, KW, _ = w.shape
        g = nat.f32c(g)
        if ch or cw:  # the rows / columns the forward did not compute carry no gradient
            full = th.zeros(N, g.shape[1] + ch, g.shape[2] + cw, Co, device=g.device,
                            dtype=th.float32)
            full[:, :g.shape[1], :g.shape[2]] = g
            g = full
        _, Ho, Wo, _ = g.shape
        g_x = g_w = g_b = None
        if has_bias and ctx.needs_input_grad[2]:
            g_b = colreduce(0, g.view(-1, Co))
        if not transposed:
            if ctx.needs_input_grad[0]:
                # conv_transpose2d(g_y, W): weight in the transposed-form layout [Ci, KH, KW, Co]
                wt = w.permute(3, 1, 2, 0).contiguous()
                oph = H - ((Ho - 1) * sh - 2 * ph + KH)
                opw = W - ((Wo - 1) * sw - 2 * pw + KW)
                with th.no_grad():
                    g_x = nn_ops.conv2d_nhwc(g, wt, None, None, stride=(sh, sw), padding=(ph, pw),
                                             transposed=True, output_padding=(oph, opw))
            if ctx.needs_input_grad[1]:
                g_w = _conv_weight_grad(x, g, KH, KW, (sh, sw), (ph, pw))
        else:
            if ctx.needs_input_grad[0]:
                wf = w.permute(3, 1, 2, 0).contiguous()  # F': Ci x KH x KW x Co
                with th.no_grad():
                    g_x = nn_ops.conv2d_nhwc(g, wf, None, None, stride=(sh, sw), padding=(ph, pw))
                if tuple(g_x.shape) != tuple(x.shape):
                    raise RuntimeError(f"transposed conv backward: {tuple(g_x.shape)} != "
                                       f"{tuple(x.shape)}")
            if ctx.needs_input_grad[1]:
                g_w = _conv_weight_grad(g, x, KH, KW, (sh, sw), (ph, pw)).permute(3, 1, 2, 0)
        return g_x, g_w, g_b, None, None, None, None, None


class DccrnMaskFn(th.autograd.Function):
    """aps_dccrn_mask (complex ratio masks / masked spectrograms of DCCRN, dccrn.py:217-242) with
    its adjoint aps_dccrn_mask_backward"""

    @staticmethod
    def forward(ctx, dec, store, S, nl, apply, cplx, eps):
        decc = _f32(dec)
        stc = None if store is None else _f32(store)
        N, T, Fd = decc.shape[:3]
        rows = N * T * Fd
        shape = (S, N, T, Fd, 2) if cplx or apply else (S, N, T, Fd)
        out = th.empty(*shape, device=decc.device, dtype=th.float32)
        rc = nat.load().aps_dccrn_mask(nat.ptr(decc), nat.ptr(stc), nat.ptr(out), rows, S, nl,
                                       int(apply), int(cplx), float(eps), nat.stream_of(decc))
        nat.check(rc, "aps_dccrn_mask")
        ctx.save_for_backward(decc, stc)
        ctx.cfg = (rows, S, nl, int(apply), int(cplx), float(eps))
        return out

    @staticmethod
    def backward(ctx, g):
        decc, stc = ctx.saved_tensors
        rows, S, nl, apply, cplx, eps = ctx.cfg
        g = nat.f32c(g)
        g_dec = th.empty_like(decc)
        want_store = apply and stc is not None and ctx.needs_input_grad[1]
        g_store = th.empty_like(stc) if want_store else None
        rc = nat.load().aps_dccrn_mask_backward(nat.ptr(decc), nat.ptr(stc if apply else None),
                                                nat.ptr(g), nat.ptr(g_dec), nat.ptr(g_store), rows,
                                                S, nl, apply, cplx, eps, nat.stream_of(decc))
        nat.check(rc, "aps_dccrn_mask_backward")
        return g_dec, g_store, None, None, None, None, None


class PosencFn(th.autograd.Function):
    """x * factor + sinusoid: linear in x"""

    @staticmethod
    def forward(ctx, x, div_term, factor, t0):
        from aps_amd import nn_ops
        ctx.factor = factor
        with th.no_grad():
            return nn_ops.posenc_add(x.detach(), div_term.detach(), factor, t0)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.factor, None, None, None


# ------------------------------------------------------------------------------------------------
# LSTM (nn.LSTM stacks, uni- or bidirectional, batch_first, zero initial state)
# ------------------------------------------------------------------------------------------------
def reverse_time(x: th.Tensor, lens: Optional[th.Tensor]) -> th.Tensor:
    """x N x T x D -> each utterance reversed inside its own length (zeros past it)"""
    xc = nat.f32c(x)
    N, T, D = xc.shape
    out = th.empty_like(xc)
    nat.check(nat.load().aps_reverse_time(nat.ptr(xc), nat.ptr(lens), nat.ptr(out), N, T, D,
                                          nat.stream_of(xc)), "aps_reverse_time")
    return out


def _lstm_direction_backward(inp, y, g_y, w_ih, w_hh, b_ih, b_hh, lens, need_inp):
    """BPTT of one forward-direction LSTM layer given its input, its output y and g_y (all
    N x T x .): gates and cells are recomputed from y, then the reverse-time sweep, then the weight
    gradients as batched GEMMs.  -> (g_inp | None, g_w_ih, g_w_hh, g_b | None)"""
    lib = nat.load()
    N, T, D = inp.shape
    H = w_hh.shape[1]
    st = nat.stream_of(inp)
    hprev = th.empty(N, T, H, device=inp.device, dtype=th.float32)
    nat.check(lib.aps_time_shift(nat.ptr(y), nat.ptr(hprev), N, T, H, st), "aps_time_shift")
    pre = _linear_nograd(inp.reshape(N * T, D), w_ih, b_ih)
    hh = _linear_nograd(hprev.view(N * T, H), w_hh)
    gates = th.empty(N, T, 4 * H, device=inp.device, dtype=th.float32)
    cells = th.empty(N, T, H, device=inp.device, dtype=th.float32)
    nat.check(lib.aps_lstm_gate_scan(nat.ptr(pre), nat.ptr(hh), nat.ptr(b_hh), nat.ptr(lens),
                                     nat.ptr(gates), nat.ptr(cells), N, T, H, st),
              "aps_lstm_gate_scan")
    del pre, hh
    g_pre = th.empty(N, T, 4 * H, device=inp.device, dtype=th.float32)
    g_h = th.empty(N, H, device=inp.device, dtype=th.float32)
    g_c = th.empty(N, H, device=inp.device, dtype=th.float32)
    w_hh_t = transpose2d(w_hh)  # [H, 4H]
    nat.check(lib.aps_lstm_backward_sweep(nat.ptr(gates), nat.ptr(cells), nat.ptr(g_y),
                                          nat.ptr(w_hh_t), nat.ptr(lens), nat.ptr(g_pre),
                                          nat.ptr(g_h), nat.ptr(g_c), N, T, H, st),
              "aps_lstm_backward_sweep")
    del gates, cells
    gp2 = g_pre.view(N * T, 4 * H)
    g_w_ih = xty(gp2, inp.reshape(N * T, D), colsum=b_ih is not None)
    g_b = None
    if b_ih is not None:
        g_w_ih, g_b = g_w_ih
    g_w_hh = xty(gp2, hprev.view(N * T, H))
    g_inp = _linear_nograd(gp2, transpose2d(w_ih)).view(N, T, D) if need_inp else None
    return g_inp, g_w_ih, g_w_hh, g_b


class FixedBeamFn(th.autograd.Function):
    """FixedBeamformer (aps/transform/enh.py:349-384): b = sum_c conj(w[beam, c, f]) x[n, c, f, t]; gradients of
    the input and -- FixedBeamformer(requires_grad=True) -- of the coefficients (aps_fixed_beamform_backward)"""

    @staticmethod
    def forward(ctx, real, imag, w_real, w_imag, sel):
        lib = nat.load()
        r, i = _f32(real), _f32(imag)
        wr, wi = _f32(w_real).reshape(w_real.shape[:3]), _f32(w_imag).reshape(w_imag.shape[:3])
        N, Cn, F, T = r.shape
        B = wr.shape[0]
        shape = (N, F, T) if sel is not None else (N, B, F, T)
        br = th.empty(*shape, device=r.device, dtype=th.float32)
        bi = th.empty(*shape, device=r.device, dtype=th.float32)
        rc = lib.aps_fixed_beamform(nat.ptr(r), nat.ptr(i), nat.ptr(wr), nat.ptr(wi), nat.ptr(sel), nat.ptr(br),
                                    nat.ptr(bi), N, Cn, F, T, B, nat.stream_of(r))
        nat.check(rc, "aps_fixed_beamform")
        ctx.save_for_backward(r, i, wr, wi, sel)
        ctx.wshape = tuple(w_real.shape)
        return br, bi

    @staticmethod
    def backward(ctx, g_br, g_bi):
        r, i, wr, wi, sel = ctx.saved_tensors
        lib = nat.load()
        N, Cn, F, T = r.shape
        B = wr.shape[0]
        gr = nat.f32c(g_br) if g_br is not None else th.zeros((N, F, T) if sel is not None else (N, B, F, T),
                                                                device=r.device)
        gi = nat.f32c(g_bi) if g_bi is not None else th.zeros_like(gr)
        want_x = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        want_w = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        g_xr = th.empty_like(r) if want_x else None
        g_xi = th.empty_like(i) if want_x else None
        g_wr = th.empty_like(wr) if want_w else None
        g_wi = th.empty_like(wi) if want_w else None
        rc = lib.aps_fixed_beamform_backward(nat.ptr(gr), nat.ptr(gi), nat.ptr(r), nat.ptr(i), nat.ptr(wr),
                                             nat.ptr(wi), nat.ptr(sel), nat.ptr(g_xr), nat.ptr(g_xi),
                                             nat.ptr(g_wr), nat.ptr(g_wi), N, Cn, F, T, B, nat.stream_of(r))
        nat.check(rc, "aps_fixed_beamform_backward")
        if want_w:
            g_wr, g_wi = g_wr.view(ctx.wshape), g_wi.view(ctx.wshape)
        return g_xr, g_xi, g_wr, g_wi, None


class RnnStepFn(th.autograd.Function):
    """One layer and direction of nn.GRU / nn.RNN (tanh | relu) / nn.LSTM (any hidden size, no projection)
    under autograd, step by step: the recurrences of var_len_rnn_forward (aps/asr/base/component.py:26-55)
    that have no persistent kernel.  forward = the loop of nn_ops.rnn_step_forward (one aps_linear + one
    aps_rnn_step per step) keeping the states; backward = BPTT with, per step, the recomputation of
    h_{t-1} W_hh^T + b_hh, one aps_rnn_step_backward and g_gh W_hh, then the batched weight / bias gradients
    (aps_gemm_tn) and g_x = g_gx W_ih.  reverse: the utterances run time-reversed inside their lengths."""

    @staticmethod
    def forward(ctx, x, lens, mode, reverse, w_ih, w_hh, b_ih, b_hh):
        lib = nat.load()
        xc = _f32(x)
        N, T, D = xc.shape
        G = {0: 3, 1: 1, 2: 1, 3: 4}[mode]
        H = w_hh.shape[1]
        if w_hh.shape[0] != G * H:
            raise NotImplementedError("aps_amd: no HIP backward for a projected LSTM (proj_size > 0)")
        st = nat.stream_of(xc)
        dev = xc.device
        inp = reverse_time(xc, lens) if reverse else xc
        wi, wh = _f32(w_ih), _f32(w_hh)
        bi = None if b_ih is None else _f32(b_ih)
        bh = None if b_hh is None else _f32(b_hh)
        gx = _linear_nograd(inp.reshape(N * T, D), wi, bi).view(N, T, G * H)
        hs = th.empty(N, T, H, device=dev, dtype=th.float32)   # the state after step t (frozen past len)
        cs = th.empty(N, T, H, device=dev, dtype=th.float32) if mode == 3 else None
        y = th.empty(N, T, H, device=dev, dtype=th.float32)
        h = th.zeros(N, H, device=dev, dtype=th.float32)
        c = th.zeros(N, H, device=dev, dtype=th.float32) if mode == 3 else None
        for t in range(T):
            gh = _linear_nograd(h, wh, bh)
            h_new = th.empty(N, H, device=dev, dtype=th.float32)
            c_new = th.empty(N, H, device=dev, dtype=th.float32) if mode == 3 else None
            rc = lib.aps_rnn_step(nat.ptr(gx[:, t]), T * G * H, nat.ptr(gh), nat.ptr(h), nat.ptr(c),
                                  nat.ptr(lens), t, nat.ptr(h_new), nat.ptr(c_new), nat.ptr(y[:, t]), T * H,
                                  N, H, mode, st)
            nat.check(rc, "aps_rnn_step")
            hs[:, t] = h_new
            if mode == 3:
                cs[:, t] = c_new
            h, c = h_new, c_new
        ctx.save_for_backward(inp, lens, wi, wh, bh, gx, hs, cs)
        ctx.cfg = (mode, bool(reverse), b_ih is not None, b_hh is not None)
        return reverse_time(y, lens) if reverse else y

    @staticmethod
    def backward(ctx, g_out):
        inp, lens, wi, wh, bh, gx, hs, cs = ctx.saved_tensors
        mode, reverse, has_bi, has_bh = ctx.cfg
        lib = nat.load()
        N, T, D = inp.shape
        H = wh.shape[1]
        G = wh.shape[0] // H
        st = nat.stream_of(inp)
        dev = inp.device
        g_y = nat.f32c(g_out)
        if reverse:
            g_y = reverse_time(g_y, lens)
        g_gx = th.empty(N, T, G * H, device=dev, dtype=th.float32)
        g_gh = th.empty(N, T, G * H, device=dev, dtype=th.float32)
        wh_t = transpose2d(wh)  # [H, G H]: g_gh W_hh as the forward GEMM
        zeros = th.zeros(N, H, device=dev, dtype=th.float32)
        g_h = g_c = None
        for t in range(T - 1, -1, -1):
            hp = hs[:, t - 1].contiguous() if t else zeros
            cp = (cs[:, t - 1].contiguous() if t else zeros) if mode == 3 else None
            gh = _linear_nograd(hp, wh, bh)
            g_hp = th.empty(N, H, device=dev, dtype=th.float32)
            g_cp = th.empty(N, H, device=dev, dtype=th.float32) if mode == 3 else None
            rc = lib.aps_rnn_step_backward(nat.ptr(gx[:, t]), T * G * H, nat.ptr(gh), nat.ptr(hp), nat.ptr(cp),
                                           nat.ptr(lens), t, nat.ptr(g_y[:, t]), T * H, nat.ptr(g_h),
                                           nat.ptr(g_c), nat.ptr(g_gx[:, t]), nat.ptr(g_gh[:, t]), T * G * H,
                                           nat.ptr(g_hp), nat.ptr(g_cp), N, H, mode, st)
            nat.check(rc, "aps_rnn_step_backward")
            if t:  # g_h_{t-1} = the direct part + g_gh_t W_hh
                g_h = act_forward(_linear_nograd(g_gh[:, t].contiguous(), wh_t), g_hp, 0, 1.0)
                g_c = g_cp
        g2x, g2h = g_gx.view(N * T, G * H), g_gh.view(N * T, G * H)
        hprev = th.zeros(N, T, H, device=dev, dtype=th.float32)
        if T > 1:
            hprev[:, 1:] = hs[:, :-1]
        g_w_ih = xty(g2x, inp.reshape(N * T, D), colsum=has_bi)
        g_b_ih = None
        if has_bi:
            g_w_ih, g_b_ih = g_w_ih
        g_w_hh = xty(g2h, hprev.view(N * T, H), colsum=has_bh)
        g_b_hh = None
        if has_bh:
            g_w_hh, g_b_hh = g_w_hh
        g_x = None
        if ctx.needs_input_grad[0]:
            g_x = _linear_nograd(g2x, transpose2d(wi)).view(N, T, D)
            if reverse:
                g_x = reverse_time(g_x, lens)
        return g_x, None, None, None, g_w_ih, g_w_hh, g_b_ih, g_b_hh


class RnnCellFn(th.autograd.Function):
    """ONE step of a GRU / tanh or ReLU RNN / LSTM cell with the state carried by the caller -- the RNN attention
    decoder's step (aps/asr/base/decoder.py:112-165), whose next input depends on this step's attention, so the
    sequence cannot be handed to RnnStepFn whole.  gx = x W_ih^T + b_ih, gh = h_prev W_hh^T + b_hh [N, G H] (both
    from `linear`, whose own adjoint carries the gradients on to x, h_prev and the weights), h_prev [N, H] (None
    for an LSTM whose recurrent vector is a projection: the cell itself never reads it), c_prev [N, H] | None
    -> (h, c | None).  forward aps_rnn_step, backward aps_rnn_step_backward: g_gx, g_gh and the DIRECT parts of
    the state gradients (the GRU's z h_prev, the LSTM's f c_prev)."""

    @staticmethod
    def forward(ctx, gx, gh, h_prev, c_prev, mode, H):
        gxc, ghc = _f32(gx), _f32(gh)
        hp = None if h_prev is None else _f32(h_prev)
        cp = None if c_prev is None else _f32(c_prev)
        N, GH = gxc.shape
        h_new = th.empty(N, H, device=gxc.device, dtype=th.float32)
        c_new = th.empty(N, H, device=gxc.device, dtype=th.float32) if mode == 3 else None
        rc = nat.load().aps_rnn_step(nat.ptr(gxc), GH, nat.ptr(ghc), nat.ptr(hp), nat.ptr(cp), nat.ptr(None), 0,
                                     nat.ptr(h_new), nat.ptr(c_new), nat.ptr(None), 0, N, H, int(mode),
                                     nat.stream_of(gxc))
        nat.check(rc, "aps_rnn_step")
        ctx.save_for_backward(gxc, ghc, hp, cp)
        ctx.cfg = (int(mode), int(H), h_prev is not None, c_prev is not None)
        return h_new, c_new

    @staticmethod
    def backward(ctx, g_h, g_c):
        gx, gh, hp, cp = ctx.saved_tensors
        mode, H, has_h, has_c = ctx.cfg
        N, GH = gx.shape
        dev = gx.device
        g_h = None if g_h is None else nat.f32c(g_h)
        g_c = None if g_c is None or mode != 3 else nat.f32c(g_c)
        g_gx = th.empty(N, GH, device=dev, dtype=th.float32)
        g_gh = th.empty(N, GH, device=dev, dtype=th.float32)
        g_hp = th.empty(N, H, device=dev, dtype=th.float32)
        g_cp = th.empty(N, H, device=dev, dtype=th.float32) if mode == 3 else None
        rc = nat.load().aps_rnn_step_backward(nat.ptr(gx), GH, nat.ptr(gh), nat.ptr(hp), nat.ptr(cp), nat.ptr(None),
                                              0, nat.ptr(None), 0, nat.ptr(g_h), nat.ptr(g_c), nat.ptr(g_gx),
                                              nat.ptr(g_gh), GH, nat.ptr(g_hp), nat.ptr(g_cp), N, H, mode,
                                              nat.stream_of(gx))
        nat.check(rc, "aps_rnn_step_backward")
        return g_gx, g_gh, (g_hp if has_h else None), (g_cp if has_c else None), None, None


class LstmProjStepFn(th.autograd.Function):
    """One layer and direction of nn.LSTM(proj_size = P > 0) under autograd, step by step: the cell of
    RnnStepFn (mode 3) with h_t = W_hr (o tanh(c_t)) emitted and fed back (torch.nn.LSTM with projections,
    what PyTorchRNN(..., proj_size=P) builds, aps/asr/base/component.py:145-190).  The recurrent state is the
    PROJECTED vector, so the packed-sequence freeze and the state gradient's pass-through act on it."""

    @staticmethod
    def forward(ctx, x, lens, reverse, w_ih, w_hh, w_hr, b_ih, b_hh):
        lib = nat.load()
        xc = _f32(x)
        N, T, D = xc.shape
        P, H = w_hr.shape
        st = nat.stream_of(xc)
        dev = xc.device
        inp = reverse_time(xc, lens) if reverse else xc
        wi, wh, wr = _f32(w_ih), _f32(w_hh), _f32(w_hr)
        bi = None if b_ih is None else _f32(b_ih)
        bh = None if b_hh is None else _f32(b_hh)
        gx = _linear_nograd(inp.reshape(N * T, D), wi, bi).view(N, T, 4 * H)
        hf = th.empty(N, T, H, device=dev, dtype=th.float32)   # o tanh(c) of step t
        ps = th.empty(N, T, P, device=dev, dtype=th.float32)   # the projected state after step t (frozen past len)
        cs = th.empty(N, T, H, device=dev, dtype=th.float32)
        y = th.zeros(N, T, P, device=dev, dtype=th.float32)
        hp = th.zeros(N, P, device=dev, dtype=th.float32)
        c = th.zeros(N, H, device=dev, dtype=th.float32)
        for t in range(T):
            gh = _linear_nograd(hp, wh, bh)
            h_new = th.empty(N, H, device=dev, dtype=th.float32)
            c_new = th.empty(N, H, device=dev, dtype=th.float32)
            rc = lib.aps_rnn_step(nat.ptr(gx[:, t]), T * 4 * H, nat.ptr(gh), None, nat.ptr(c), nat.ptr(lens), t,
                                  nat.ptr(h_new), nat.ptr(c_new), None, 0, N, H, 3, st)
            nat.check(rc, "aps_rnn_step")
            proj = _linear_nograd(h_new, wr)
            if lens is not None:
                live = (lens > t)[:, None]
                proj = th.where(live, proj, hp)
                y[:, t] = th.where(live, proj, th.zeros_like(proj))
            else:
                y[:, t] = proj
            hf[:, t], ps[:, t], cs[:, t] = h_new, proj, c_new
            hp, c = proj, c_new
        ctx.save_for_backward(inp, lens, wi, wh, wr, bh, gx, hf, ps, cs)
        ctx.cfg = (bool(reverse), b_ih is not None, b_hh is not None)
        return reverse_time(y, lens) if reverse else y

    @staticmethod
    def backward(ctx, g_out):
        inp, lens, wi, wh, wr, bh, gx, hf, ps, cs = ctx.saved_tensors
        reverse, has_bi, has_bh = ctx.cfg
        lib = nat.load()
        N, T, D = inp.shape
        P, H = wr.shape
        st = nat.stream_of(inp)
        dev = inp.device
        g_y = nat.f32c(g_out)
        if reverse:
            g_y = reverse_time(g_y, lens)
        g_gx = th.empty(N, T, 4 * H, device=dev, dtype=th.float32)
        g_gh = th.empty(N, T, 4 * H, device=dev, dtype=th.float32)
        g_ps = th.zeros(N, T, P, device=dev, dtype=th.float32)  # gradient of the projection's OUTPUT, live rows
        wh_t, wr_t = transpose2d(wh), transpose2d(wr)  # [P, 4H], [H, P]
        zp = th.zeros(N, P, device=dev, dtype=th.float32)
        zh = th.zeros(N, H, device=dev, dtype=th.float32)
        g_p = g_c = None   # carried from t + 1
        for t in range(T - 1, -1, -1):
            live = None if lens is None else (lens > t)[:, None]
            g_here = g_y[:, t] if g_p is None else g_y[:, t] + g_p
            if live is not None:  # a frozen row emitted zeros: only the carried part exists, and it passes through
                g_here = th.where(live, g_here, zp)
            g_ps[:, t] = g_here
            g_hf = _linear_nograd(g_here.contiguous(), wr_t)  # [N, H]: through h = W_hr (o tanh c)
            hp = ps[:, t - 1].contiguous() if t else zp
            cp = cs[:, t - 1].contiguous() if t else zh
            gh = _linear_nograd(hp, wh, bh)
            g_hp_unused = th.empty(N, H, device=dev, dtype=th.float32)
            g_cp = th.empty(N, H, device=dev, dtype=th.float32)
            rc = lib.aps_rnn_step_backward(nat.ptr(gx[:, t]), T * 4 * H, nat.ptr(gh), None, nat.ptr(cp),
                                           nat.ptr(lens), t, None, 0, nat.ptr(g_hf), nat.ptr(g_c),
                                           nat.ptr(g_gx[:, t]), nat.ptr(g_gh[:, t]), T * 4 * H,
                                           nat.ptr(g_hp_unused), nat.ptr(g_cp), N, H, 3, st)
            nat.check(rc, "aps_rnn_step_backward")
            if t:
                nxt = _linear_nograd(g_gh[:, t].contiguous(), wh_t)  # [N, P]
                if live is not None and g_p is not None:
                    nxt = nxt + th.where(live, zp, g_p)   # frozen rows hand their carried gradient on
                g_p, g_c = nxt, g_cp
        g2x, g2h = g_gx.view(N * T, 4 * H), g_gh.view(N * T, 4 * H)
        pprev = th.zeros(N, T, P, device=dev, dtype=th.float32)
        if T > 1:
            pprev[:, 1:] = ps[:, :-1]
        g_w_ih = xty(g2x, inp.reshape(N * T, D), colsum=has_bi)
        g_b_ih = None
        if has_bi:
            g_w_ih, g_b_ih = g_w_ih
        g_w_hh = xty(g2h, pprev.view(N * T, P), colsum=has_bh)
        g_b_hh = None
        if has_bh:
            g_w_hh, g_b_hh = g_w_hh
        g_w_hr = xty(g_ps.view(N * T, P), hf.view(N * T, H))
        g_x = None
        if ctx.needs_input_grad[0]:
            g_x = _linear_nograd(g2x, transpose2d(wi)).view(N, T, D)
            if reverse:
                g_x = reverse_time(g_x, lens)
        return g_x, None, None, g_w_ih, g_w_hh, g_w_hr, g_b_ih, g_b_hh


class LstmFn(th.autograd.Function):
    """forward: the persistent recurrence kernels (aps_lstm_stack / aps_lstm_layer); backward: per
    layer and direction the BPTT of `_lstm_direction_backward`; the backward direction of a
    bidirectional layer runs it on time-reversed utterances (aps_reverse_time).
    flat = per layer [and direction] (w_ih, w_hh[, b_ih, b_hh])."""

    @staticmethod
    def forward(ctx, x, lens, num_layers, has_bias, bidirectional, *flat):
        from aps_amd import nn_ops
        per = 4 if has_bias else 2
        dirs = 2 if bidirectional else 1
        sets = [tuple(_f32(t) for t in flat[i * per:(i + 1) * per])
                for i in range(num_layers * dirs)]
        with th.no_grad():
            if bidirectional:
                ys = nn_ops.lstm_bidir_layers_forward(
                    [(sets[2 * l], sets[2 * l + 1]) for l in range(num_layers)], _f32(x), lens,
                    has_bias)
            else:
                ys = nn_ops.lstm_layers_forward(sets, _f32(x), lens, has_bias)
        ctx.save_for_backward(_f32(x), lens, *[t for st in sets for t in st], *ys)
        ctx.cfg = (num_layers, has_bias, dirs, len(flat))
        return ys[-1]

    @staticmethod
    def backward(ctx, g):
        L, has_bias, dirs, nflat = ctx.cfg
        saved = ctx.saved_tensors
        x, lens = saved[0], saved[1]
        flat, ys = saved[2:2 + nflat], saved[2 + nflat:]
        per = 4 if has_bias else 2
        H = flat[1].shape[1]
        g_y = nat.f32c(g)
        grads = [None] * nflat
        for l in range(L - 1, -1, -1):
            inp = x if l == 0 else ys[l - 1]
            need_inp = l > 0 or ctx.needs_input_grad[0]
            g_inp = None
            for d in range(dirs):
                base = (l * dirs + d) * per
                w_ih, w_hh = flat[base], flat[base + 1]
                b_ih = flat[base + 2] if has_bias else None
                b_hh = flat[base + 3] if has_bias else None
                if dirs == 1:
                    y_d, g_d, inp_d = ys[l], g_y, inp
                else:
                    y_d = ys[l][..., d * H:(d + 1) * H].contiguous()
                    g_d = g_y[..., d * H:(d + 1) * H].contiguous()
                    inp_d = inp
                    if d == 1:  # the backward direction: the same sweep on reversed utterances
                        y_d, g_d = reverse_time(y_d, lens), reverse_time(g_d, lens)
                        inp_d = reverse_time(inp, lens)
                gi, gw_ih, gw_hh, gb = _lstm_direction_backward(inp_d, y_d, g_d, w_ih, w_hh, b_ih,
                                                                b_hh, lens, need_inp)
                grads[base], grads[base + 1] = gw_ih, gw_hh
                if has_bias:
                    grads[base + 2], grads[base + 3] = gb, gb.clone()
                if gi is not None:
                    if d == 1:
                        gi = reverse_time(gi, lens)
                    g_inp = gi if g_inp is None else act_forward(gi, g_inp, 0, 1.0)  # sum
            g_y = g_inp
        return (g_y, None, None, None, None) + tuple(grads)


# ------------------------------------------------------------------------------------------------
# mask-based MVDR (aps/asr/filter/mvdr.py:29-174)
# ------------------------------------------------------------------------------------------------
class CovarianceFn(th.autograd.Function):
    """(Rs, Rn) from the raw masks; gradients to the masks (the spectrogram is data).  mask_n None: the
    reference's implicit noise mask, Rn = estimate_covar(1 - m', X) with m' the PROCESSED speech mask
    (aps/asr/filter/mvdr.py:131-135) -- Rn's gradient w.r.t. the complement (over all T frames, no
    processing of its own) is subtracted from the processed speech mask's inside the speech branch's
    adjoint, in front of _process_mask's own"""

    @staticmethod
    def forward(ctx, store, mask_s, mask_n, x_len, mask_norm):
        from aps_amd.asr.filter import mvdr as M
        ms = _f32(mask_s)
        mn = None if mask_n is None else _f32(mask_n)
        with th.no_grad():
            cov_s, cov_n = M.covariance(store.detach(), ms, mn, x_len, mask_norm=mask_norm)
        ctx.save_for_backward(store.detach(), ms, mn, x_len, cov_s, cov_n)
        ctx.mask_norm = mask_norm
        return cov_s, cov_n

    @staticmethod
    def backward(ctx, g_s, g_n):
        store, ms, mn, x_len, cov_s, cov_n = ctx.saved_tensors
        lib = nat.load()
        N, Cn, T, F, _ = store.shape

        def adjoint(mask, lens, cov, g, norm, g_sub=None):
            g_mask = th.empty_like(mask)
            rc = lib.aps_mvdr_covariance_backward(nat.ptr(store), nat.ptr(mask), nat.ptr(lens),
                                                  nat.ptr(cov), nat.ptr(nat.f32c(g)),
                                                  nat.ptr(g_mask), N, Cn, T, F, store.stride(0),
                                                  store.stride(1), store.stride(2), int(norm),
                                                  nat.ptr(g_sub), nat.stream_of(store))
            nat.check(rc, "aps_mvdr_covariance_backward")
            return g_mask

        if mn is not None:
            return None, adjoint(ms, x_len, cov_s, g_s, ctx.mask_norm), \
                adjoint(mn, x_len, cov_n, g_n, ctx.mask_norm), None, None
        comp = th.empty_like(ms)  # 1 - m', N x T x F
        rc = lib.aps_mvdr_process_mask(nat.ptr(ms), nat.ptr(x_len), N, T, F, int(ctx.mask_norm), 1,
                                       nat.ptr(comp), nat.stream_of(ms))
        nat.check(rc, "aps_mvdr_process_mask")
        g_comp = adjoint(comp, None, cov_n, g_n, 0)
        return None, adjoint(ms, x_len, cov_s, g_s, ctx.mask_norm, g_comp), None, None, None


class OffdiagAbsFn(th.autograd.Function):
    """Rs N x F x C x C x 2 -> |off-diagonal row mean| N x C x F (mvdr.py:165-170)"""

    @staticmethod
    def forward(ctx, cov):
        c = _f32(cov)
        N, F, Cn = c.shape[:3]
        v = th.empty(N, Cn, F, device=c.device, dtype=th.float32)
        nat.check(nat.load().aps_mvdr_offdiag_abs(nat.ptr(c), nat.ptr(v), N, Cn, F,
                                                  nat.stream_of(c)), "aps_mvdr_offdiag_abs")
        ctx.save_for_backward(c)
        return v

    @staticmethod
    def backward(ctx, g):
        (c,) = ctx.saved_tensors
        N, F, Cn = c.shape[:3]
        g_cov = th.zeros_like(c)
        nat.check(nat.load().aps_mvdr_offdiag_abs_backward(nat.ptr(c), nat.ptr(nat.f32c(g)),
                                                           nat.ptr(g_cov), N, Cn, F,
                                                           nat.stream_of(c)),
                  "aps_mvdr_offdiag_abs_backward")
        return g_cov


class SoftmaxRowsFn(th.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xc = _f32(x)
        D = xc.shape[-1]
        y = th.empty_like(xc)
        nat.check(nat.load().aps_softmax_rows(nat.ptr(xc), nat.ptr(y), xc.numel() // D, D,
                                              nat.stream_of(xc)), "aps_softmax_rows")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        D = y.shape[-1]
        g_x = th.empty_like(y)
        nat.check(nat.load().aps_softmax_rows_backward(nat.ptr(y), nat.ptr(nat.f32c(g)),
                                                       nat.ptr(g_x), y.numel() // D, D,
                                                       nat.stream_of(y)),
                  "aps_softmax_rows_backward")
        return g_x


class WeightFn(th.autograd.Function):
    """w = (Rn + eps I)^-1 Rs u / (tr(.) + eps) (mvdr.py:75-101)"""

    @staticmethod
    def forward(ctx, cov_s, cov_n, u, eps):
        cs, cn, uu = _f32(cov_s), _f32(cov_n), _f32(u)
        N, F, Cn = cs.shape[:3]
        w = th.empty(N, F, Cn, 2, device=cs.device, dtype=th.float32)
        from aps_amd.ops import mvdr_singular_check, mvdr_singular_flag
        rc = nat.load().aps_mvdr_weight(nat.ptr(cs), nat.ptr(cn), nat.ptr(uu), N, Cn, F, float(eps),
                                        nat.ptr(w), nat.ptr(mvdr_singular_flag(cs.device)), nat.stream_of(cs))
        nat.check(rc, "aps_mvdr_weight")
        mvdr_singular_check("deferred", tuple(cn.shape), cn.device)
        ctx.save_for_backward(cs, cn, uu)
        ctx.eps = eps
        return w

    @staticmethod
    def backward(ctx, g):
        cs, cn, uu = ctx.saved_tensors
        N, F, Cn = cs.shape[:3]
        g_s, g_n = th.empty_like(cs), th.empty_like(cn)
        part = th.empty(N, F, Cn, device=cs.device, dtype=th.float32)
        rc = nat.load().aps_mvdr_weight_backward(nat.ptr(cs), nat.ptr(cn), nat.ptr(uu),
                                                 nat.ptr(nat.f32c(g)), nat.ptr(g_s), nat.ptr(g_n),
                                                 nat.ptr(part), N, Cn, F, float(ctx.eps),
                                                 nat.stream_of(cs))
        nat.check(rc, "aps_mvdr_weight_backward")
        # g_u[n, c] = sum_f part[n, f, c]: ONE column reduction over the bins of the [F, N C] view
        # (a strided copy of N F C floats in front of it instead of N launches and N workspaces)
        g_u = colreduce(0, part.permute(1, 0, 2).reshape(F, N * Cn)).view(N, Cn)
        return g_s, g_n, g_u, None


class BeamformFn(th.autograd.Function):
    """y = sum_c conj(w_c) x_c (mvdr.py:29-39); gradient to the weights"""

    @staticmethod
    def forward(ctx, store, weight):
        from aps_amd.asr.filter import mvdr as M
        w = _f32(weight)
        with th.no_grad():
            y = M.beamform_store(store.detach(), w)
        ctx.save_for_backward(store.detach())
        return y

    @staticmethod
    def backward(ctx, g):
        (store,) = ctx.saved_tensors
        N, Cn, T, F, _ = store.shape
        g_w = th.empty(N, F, Cn, 2, device=store.device, dtype=th.float32)
        rc = nat.load().aps_mvdr_beamform_backward(nat.ptr(store), nat.ptr(nat.f32c(g)),
                                                   nat.ptr(g_w), N, Cn, T, F, store.stride(0),
                                                   store.stride(1), store.stride(2),
                                                   nat.stream_of(store))
        nat.check(rc, "aps_mvdr_beamform_backward")
        return None, g_w


# ------------------------------------------------------------------------------------------------
# AsrTransform("abs-mel-log-cmvn") on the beamformer output (asr.py:306-332, 360-464, 576-618)
# ------------------------------------------------------------------------------------------------
class MagnitudeFn(th.autograd.Function):
    """|z + eps| of interleaved complex values (..., 2) -> (...)"""

    @staticmethod
    def forward(ctx, z, eps):
        zc = _f32(z)
        out = th.empty(zc.shape[:-1], device=zc.device, dtype=th.float32)
        rc = nat.load().aps_magnitude_forward(nat.ptr(zc), nat.ptr(out), out.numel(), float(eps),
                                              nat.stream_of(zc))
        nat.check(rc, "aps_magnitude_forward")
        ctx.save_for_backward(zc)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, g):
        (zc,) = ctx.saved_tensors
        g_z = th.empty_like(zc)
        rc = nat.load().aps_magnitude_backward(nat.ptr(zc), nat.ptr(nat.f32c(g)), nat.ptr(g_z),
                                               zc.numel() // 2, float(ctx.eps), nat.stream_of(zc))
        nat.check(rc, "aps_magnitude_backward")
        return g_z, None


class LogCmvnFn(th.autograd.Function):
    """[log] -> per-row CMVN of real rows (aps_row_features without power / mel)"""

    @staticmethod
    def forward(ctx, m, plan):
        from aps_amd import ops
        mc = _f32(m)
        with th.no_grad():
            out = ops.row_features(mc, plan)
        ctx.save_for_backward(mc)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, g):
        (mc,) = ctx.saved_tensors
        p = ctx.plan
        D = mc.shape[-1]
        g_m = th.empty_like(mc)
        rc = nat.load().aps_log_cmvn_backward(nat.ptr(mc), nat.ptr(nat.f32c(g)), nat.ptr(g_m),
                                              mc.numel() // D, D, int(p.apply_log), int(p.norm_mean),
                                              int(p.norm_var), float(p.log_eps),
                                              float(p.log_lower_bound), float(p.cmvn_eps),
                                              nat.stream_of(mc))
        nat.check(rc, "aps_log_cmvn_backward")
        return g_m, None
