#!/usr/bin/env python
"""Launch time against K at fixed M, N: slope = cost of a K step, intercept = prologue + epilogue.
   python scripts/split_k_sweep.py [M] [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd import nn_ops  # noqa: E402
from scripts.r02_probe import graph_time  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8064
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
variants = [("fp32", "0", None, None), ("bd", "1", None, "bd"), ("pc", "1", "128", "pc"), ("v1-64", "1", "64", "0"), ("v1-128", "1", "128", "0"),
            ("swp64", "1", "64", "1"), ("swp128", "1", "128", "1")]
only = os.environ.get("SPLIT_BENCH_ONLY")
torch.manual_seed(0)
with torch.no_grad():
    for tag, mode, tn, swp in variants:
        if only and tag not in only.split(","):
            continue
        nn_ops.SPLIT_MODE = mode
        nn_ops.SPLIT_LAYOUT = 1 if swp == "bd" else 0
        if tn:
            os.environ["APS_SPLIT_TN"] = tn
            os.environ["APS_SPLIT_KERNEL"] = "pc" if swp == "pc" else ("swp" if swp == "1" else "v1")
        for res in (False, True):
            row = []
            for K in (128, 256, 512, 1024, 2048):
                x = torch.randn(M, K, device="cuda")
                w = torch.nn.Parameter(torch.randn(N, K, device="cuda") / K**0.5, requires_grad=False)
                b = torch.randn(N, device="cuda")
                r = torch.randn(M, N, device="cuda") if res else None
                row.append(graph_time(lambda: nn_ops.linear(x, w, b, residual=r)))
            step = (row[-1] - row[-2]) / 32  # us per K step of 32
            print(f"[{tag:7s}] M={M} N={N} res={int(res)} | " +
                  " ".join(f"K={k}: {t:6.1f}" for k, t in zip((128, 256, 512, 1024, 2048), row)) +
                  f" | per 32-step {step * 1e3:6.0f} ns, intercept {row[-1] - 64 * step:5.1f} us",
                  flush=True)
