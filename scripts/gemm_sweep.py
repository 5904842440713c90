#!/usr/bin/env python
"""Time aps_linear per problem shape and tile shape (APS_GEMM_TILE = 1: 128x128, 2: 128x64,
3: 64x64, 0: the launcher's own choice).  Usage: python scripts/gemm_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd.nn_ops import linear  # noqa: E402

TILES = (0, 3, 5)  # 0: launcher's choice, 3: 64x64 BK32, 4: 64x64 BK64, 5: 64x64 BK32 with 2 in-workgroup K groups (1: 128x128, 2: 128x64)
SHAPES = [  # (M, N, K)
    (2016, 512, 512), (2016, 1024, 512), (2016, 512, 1024), (2016, 1536, 512), (2016, 5000, 512),
    (2016, 512, 2560), (7968, 2048, 512), (7968, 512, 1028), (7968, 514, 512), (12800, 512, 512),
    (12800, 2048, 512), (12800, 512, 2048), (12800, 1536, 512), (12800, 512, 5120),
    (4096, 4096, 4096)]


COLD = "--cold" in sys.argv  # weights streamed from HBM (a pool larger than L2 + MALL is cycled)


def bench(M, N, K, reps=30):
    x = torch.randn(M, K, device="cuda")
    copies = max(1, min(reps, int(600e6 // (N * K * 4)))) if COLD else 1
    ws = [torch.randn(N, K, device="cuda") for _ in range(copies)]
    b = torch.randn(N, device="cuda")
    for i in range(3):
        linear(x, ws[i % copies], b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(reps):
        linear(x, ws[i % copies], b)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


with torch.no_grad():
    print(f"{'M':>6} {'N':>5} {'K':>5} | " + " | ".join(f"tile{t}: us    TF" for t in TILES))
    for M, N, K in SHAPES:
        cells = []
        for t in TILES:
            os.environ["APS_GEMM_TILE"] = str(t)
            us = bench(M, N, K)
            cells.append(f"{us:9.1f} {2.0 * M * N * K / us / 1e6:5.1f}")
        print(f"{M:6d} {N:5d} {K:5d} | " + " | ".join(cells))
