#!/usr/bin/env python
"""Do GEMM launches of two streams overlap?  One graph of 40 projections (M = 2016, the 32-utterance
shapes) per stream, replayed on 1 / 2 / 3 / 4 streams at once: time per replay round.  If the launches of
different streams shared the chip, R streams would take about as long as one.
    python scripts/stream_overlap_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aps_amd import nn_ops  # noqa: E402

dev = torch.device("cuda:0")
nn_ops.SPLIT_MODE = "1"
nn_ops.PANEL_FORM = int(sys.argv[1]) if len(sys.argv) > 1 else 0
print("panel form", nn_ops.PANEL_FORM)
for (M, N, K) in ((2016, 512, 512), (2016, 1024, 512), (2016, 1536, 512)):
    g = torch.Generator().manual_seed(N)
    R = 4
    ws = [torch.nn.Parameter((torch.randn(N, K, generator=g) / K**0.5).to(dev), requires_grad=False) for _ in range(R)]
    xs = [[torch.randn(M, K, generator=g).to(dev) for _ in range(4)] for _ in range(R)]
    streams = [torch.cuda.Stream() for _ in range(R)]
    graphs = []
    with torch.no_grad():
        for r in range(R):
            def run(r=r):
                for i in range(40):
                    nn_ops.linear(xs[r][i % 4], ws[r])
            run()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                run()
            graphs.append(gr)
        line = []
        for n in (1, 2, 3, 4):
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    for r in range(n):
                        with torch.cuda.stream(streams[r]):
                            graphs[r].replay()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 5)
            line.append(f"{n} stream(s): {best * 1e6 / 40:6.1f} us per launch slot ({best * 1e6 / 40 / n:5.1f} per GEMM)")
    print(f"M={M} N={N} K={K}: " + "   ".join(line), flush=True)
