#!/usr/bin/env python
"""Does a CU-masked stream confine kernels, and what does a GEMM launch of the 32-utterance step cost on a part of
the chip?  40 launches per shape inside one captured graph replayed on (a) a plain stream, (b) a stream masked to
1/2 of the CUs, (c) 1/4; then two half-chip streams side by side against two plain streams.
    python scripts/cu_mask_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aps_amd import nn_ops, replicas  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(2016, 1024, 512, True, "swish", False), (2016, 512, 1024, False, None, True),
          (2016, 1536, 512, True, None, False), (2016, 512, 512, False, None, True)]
nn_ops.push_lstm_share(2)   # (four-wave tiles, as two batches in flight run them)


def make(shape, stream):
    M, N, K, ln, act, res = shape
    g = torch.Generator().manual_seed(M + N + K)
    xs = [torch.randn(M, K, generator=g).to(dev) for _ in range(8)]
    w = torch.nn.Parameter((torch.randn(N, K, generator=g) / K**0.5).to(dev), requires_grad=False)
    b = torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N, generator=g).to(dev) if res else None
    norm = torch.nn.LayerNorm(K).to(dev) if ln else None

    def run():
        for i in range(40):
            nn_ops.linear(xs[i % 8], w, b, r, act=act, alpha=0.5 if ln else 1.0, ln=norm)
    with torch.no_grad():
        run()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            run()
    return graph


def timed(graphs_streams, reps=5):
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        evs = []
        for graph, st in graphs_streams:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(st):
                e0.record()
                for _ in range(reps):
                    graph.replay()
                e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        best = min(best, max(a.elapsed_time(b) for a, b in evs) * 1e3 / (40 * reps))
    return best


plain = [torch.cuda.Stream(), torch.cuda.Stream()]
half = replicas.replica_streams(dev, 2, cu_split=True)
quarter = replicas.replica_streams(dev, 4, cu_split=True)
print("us per launch:   plain   half-chip   quarter-chip | two plain streams   two half-chip streams (per launch of each)")
for shape in SHAPES:
    gp = [make(shape, s) for s in plain]
    gh = [make(shape, s) for s in half]
    gq = make(shape, quarter[0])
    a = timed([(gp[0], plain[0])])
    b = timed([(gh[0], half[0])])
    c = timed([(gq, quarter[0])])
    d = timed([(gp[0], plain[0]), (gp[1], plain[1])])
    e = timed([(gh[0], half[0]), (gh[1], half[1])])
    print(f"M={shape[0]} N={shape[1]:5d} K={shape[2]:5d}   {a:6.1f}   {b:6.1f}   {c:6.1f}   |   {d:6.1f}   {e:6.1f}", flush=True)
