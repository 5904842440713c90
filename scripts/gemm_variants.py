#!/usr/bin/env python
"""Device time per GEMM launch (20 launches in a captured graph) at the merged-batch conformer
shapes; run under APS_GEMM_SWP / APS_GEMM_SWP_MAXM / APS_GEMM_TILE to compare kernel variants on
one box.   python scripts/gemm_variants.py [tag]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd.nn_ops import linear  # noqa: E402
from scripts.r02_probe import graph_time  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "default"
torch.manual_seed(0)
with torch.no_grad():
    for M in (4032, 8064):
        row = []
        for (N, K) in [(512, 512), (1024, 512), (512, 1024), (1536, 512)]:
            x = torch.randn(M, K, device="cuda")
            w = torch.randn(N, K, device="cuda") / K**0.5
            b = torch.randn(N, device="cuda")
            r = torch.randn(M, N, device="cuda")
            us = graph_time(lambda: linear(x, w, b, residual=r))
            row.append(f"{N}x{K}: {us:6.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF")
        print(f"[{tag}] M={M} | " + " | ".join(row), flush=True)
