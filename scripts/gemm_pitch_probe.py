#!/usr/bin/env python
"""Does the row pitch of the GEMM operands matter (power-of-two pitches vs padded ones)?
20 launches in a graph per shape; A / W given as strided views with pitch K + pad floats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd.nn_ops import linear  # noqa: E402

with torch.no_grad():
    torch.manual_seed(0)
    for M, N, K in [(2016, 512, 512), (2016, 1024, 512), (2016, 512, 1024), (12800, 2048, 512),
                    (12800, 512, 2048)]:
        row = []
        for pad_a, pad_w in [(0, 0), (32, 0), (64, 0), (16, 0)]:
            xa = torch.randn(M, K + pad_a, device="cuda")
            wa = torch.randn(N, K + pad_w, device="cuda") / K**0.5
            x, w = xa[:, :K], wa[:, :K]
            b = torch.randn(N, device="cuda")
            y = linear(x, w, b)
            ref = x.double() @ w.double().t() + b.double()
            err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                for _ in range(20):
                    linear(x, w, b)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            row.append(f"pad {pad_a}/{pad_w}: {e0.elapsed_time(e1) / 100 * 1e3:6.1f} us ({err:.0e})")
        print(f"{M} x {N} x {K}: " + " | ".join(row))
