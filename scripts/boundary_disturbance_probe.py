#!/usr/bin/env python
"""Do kernel BOUNDARIES of one stream slow the kernels of another?  (Round 6: the pipelined joint step takes 2.05 - 2.10 ms
whatever the stage placement, the panel form or the lookahead -- is it the ~180 dependent launches per step themselves?)
A long panel GEMM (M = 64 512 = 32 x 2016 rows, N = K = 512) on stream A, timed with events, while stream B runs
    nothing | ONE long spin kernel | a graph of 400 dependent tiny kernels (1 workgroup each) | the same tiny kernels as
    panel GEMMs of 8 rows (real kernels with cache traffic)
and the reverse view: how long the tiny chain takes alone and beside the GEMMs."""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aps_amd import nn_ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, N, K = 64512, 512, 512
x = torch.randn(M, K, generator=g).to(dev)
w = torch.nn.Parameter((torch.randn(N, K, generator=g) / K ** 0.5).to(dev), requires_grad=False)
xs = torch.randn(8, K, generator=g).to(dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
tiny = torch.zeros(64, device=dev)


def chain_tiny(n):
    for _ in range(n):
        tiny.add_(1.0)


def chain_small_gemm(n):
    for _ in range(n):
        nn_ops.linear(xs, w)


def graph_of(fn, n, stream):
    fn(2)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=stream, capture_error_mode="thread_local"):
        fn(n)
    return gr


with torch.no_grad():
    for _ in range(3):
        nn_ops.linear(x, w)
    torch.cuda.synchronize()
    big = graph_of(lambda n: [nn_ops.linear(x, w) for _ in range(n)], 8, sa)
    g_tiny = graph_of(chain_tiny, 400, sb)
    g_small = graph_of(chain_small_gemm, 400, sb)

    def timed(gr, stream, reps=3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            e0.record(stream)
            for _ in range(reps):
                gr.replay()
            e1.record(stream)
        return e0, e1, reps

    def run(other):
        torch.cuda.synchronize()
        marks_b = None
        if other == "spin":
            with torch.cuda.stream(sb):
                torch.cuda._sleep(int(2.0e7))
        elif other == "tiny":
            marks_b = timed(g_tiny, sb, 6)
        elif other == "small_gemm":
            marks_b = timed(g_small, sb, 6)
        a = timed(big, sa, 3)
        torch.cuda.synchronize()
        per_gemm = a[0].elapsed_time(a[1]) / (a[2] * 8) * 1e3
        msg = f"  stream B: {other:11s} -> the long GEMM {per_gemm:8.1f} us per launch"
        if marks_b is not None:
            msg += f"; B's chain {marks_b[0].elapsed_time(marks_b[1]) / (marks_b[2] * 400) * 1e3:6.2f} us per tiny launch"
        print(msg)

    for rnd in range(2):
        print(f"round {rnd}")
        for other in ("nothing", "spin", "tiny", "small_gemm"):
            run(other)
    for name, gr in (("tiny", g_tiny), ("small_gemm", g_small)):
        b = timed(gr, sb, 6)
        torch.cuda.synchronize()
        print(f"  B's chain alone ({name}): {b[0].elapsed_time(b[1]) / (6 * 400) * 1e3:6.2f} us per launch")
