#!/usr/bin/env python
"""Target of the rocprofv3 --pmc passes on the GEMM: 30 eager launches per shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd.nn_ops import linear  # noqa: E402

torch.manual_seed(0)
with torch.no_grad():
    for (M, N, K) in [(8064, 512, 512), (8064, 1024, 512), (8064, 512, 1024), (2016, 512, 512)]:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K**0.5
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda")
        for _ in range(30):
            linear(x, w, b, residual=r)
        torch.cuda.synchronize()
