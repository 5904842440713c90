#!/usr/bin/env python
"""A few launches of one GEMM variant for the PMC passes:
   python scripts/split_pmc_target.py <split 0|1> <M> <N> <K> [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd import nn_ops  # noqa: E402

nn_ops.SPLIT_MODE = sys.argv[1]
nn_ops.SPLIT_LAYOUT = int(os.environ.get("APS_GEMM_SPLIT_LAYOUT", "1"))
M, N, K = (int(v) for v in sys.argv[2:5])
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 10
torch.manual_seed(0)
with torch.no_grad():
    x = torch.randn(M, K, device="cuda")
    w = torch.nn.Parameter(torch.randn(N, K, device="cuda") / K**0.5, requires_grad=False)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda")
    for _ in range(launches):
        nn_ops.linear(x, w, b, residual=r, alpha=0.5)
    torch.cuda.synchronize()
