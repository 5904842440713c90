#!/usr/bin/env python
"""Device time per GEMM in a captured dependent chain as a function of K (fixed cost vs K loop)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd.nn_ops import linear  # noqa: E402

with torch.no_grad():
    for M, N in ((2016, 512), (2016, 1024)):
        line = []
        for K in (32, 64, 128, 256, 512, 1024):
            x = torch.randn(M, K, device="cuda")
            w = torch.randn(N, K, device="cuda")
            w2 = torch.randn(K, N, device="cuda")
            b = torch.randn(N, device="cuda")
            r = torch.randn(M, N, device="cuda")
            linear(linear(x, w, b, residual=r), w2)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                y = x
                for _ in range(10):
                    y = linear(linear(y, w, b, residual=r), w2)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            line.append(f"K={K}: {e0.elapsed_time(e1) / 100 * 1e3:5.1f}")
        print(f"{M}x{N}xK + {M}xKx{N} pair us:", "  ".join(line))
