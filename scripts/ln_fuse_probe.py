#!/usr/bin/env python
"""Device time of GEMM / LayerNorm + GEMM / fused LayerNorm-GEMM (graph-captured chains)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd.nn_ops import layernorm, linear  # noqa: E402


def chain_time(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


with torch.no_grad():
    for M, N, K in ((2016, 1024, 512), (2016, 1536, 512), (2016, 512, 512)):
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K**0.5
        b = torch.randn(N, device="cuda")
        ln = torch.nn.LayerNorm(K).cuda()
        t0 = chain_time(lambda: linear(x, w, b))
        t1 = chain_time(lambda: linear(layernorm(x, ln.weight, ln.bias, ln.eps), w, b))
        t2 = chain_time(lambda: linear(x, w, b, ln=ln))
        print(f"{M}x{N}x{K}: gemm {t0:6.1f} us | LN + gemm {t1:6.1f} us | fused {t2:6.1f} us")
