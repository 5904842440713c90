#!/usr/bin/env python
"""Per-launch time of the projections' GEMM at the joint step's shapes, by kernel form:
  layout 2 = aps_linear_fp16x2 (planes pass + 64 x 128 / 64 x 64 tiles), layout 3 = aps_linear_panel.
Each shape: 40 launches on 8 rotating inputs inside one captured graph (back to back, no host), HIP
events around 5 replays, best of 3; the LayerNorm-fold / residual / activation of the call sites.
    python scripts/panel_gemm_probe.py [layouts, e.g. 2,3]      (APS_PANEL_ROWS=32|64 forces a panel height)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aps_amd import nn_ops  # noqa: E402

dev = torch.device("cuda:0")
layouts = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2,31,32,33").split(",")]
# (M, N, K, ln, act, residual): conformer layer at 32 / 128 utterances, mask estimator, CTC head
SHAPES = []
for M in (2016, 8064):
    SHAPES += [(M, 1024, 512, True, "swish", False), (M, 512, 1024, False, None, True),
               (M, 1536, 512, True, None, False), (M, 512, 512, False, None, True),
               (M, 5000, 512, False, None, False)]
SHAPES += [(7968, 512, 1028, False, "relu", False), (7968, 2048, 512, False, None, False),
           (7968, 514, 512, False, "sigmoid", False), (31872, 2048, 512, False, None, False)]


def bench_shape(M, N, K, ln, act, res):
    g = torch.Generator().manual_seed(M + N + K)
    xs = [torch.randn(M, K, generator=g).to(dev) for _ in range(8)]
    w = torch.nn.Parameter((torch.randn(N, K, generator=g) / K**0.5).to(dev), requires_grad=False)
    b = torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N, generator=g).to(dev) if res else None
    norm = torch.nn.LayerNorm(K).to(dev) if ln else None
    if norm is not None:
        for p in norm.parameters():
            p.requires_grad_(False)

    def run():
        out = None
        for i in range(40):
            out = nn_ops.linear(xs[i % 8], w, b, r, act=act, alpha=0.5 if ln else 1.0, ln=norm)
        return out

    with torch.no_grad():
        ref = run()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 200)
    return best, ref


nn_ops.SPLIT_MODE = "1"
# layout 2 = planes pass + gemm_fp16x2_kernel; 3 = panel (default form); 31 | 32 | 33 = panel forms a | c | e;
# 34 | 35 = the K-group forms (16 | 8 waves, round 5)
print(f"layouts {layouts} (us per launch, executed TFLOP/s, fraction of the f16 pipe)")
for shape in SHAPES:
    M, N, K, ln, act, res = shape
    line, outs = [], []
    for lay in layouts:
        nn_ops.SPLIT_LAYOUT, nn_ops.PANEL_FORM = (3, lay - 30) if lay > 30 else (lay, 0)
        us, out = bench_shape(*shape)
        outs.append(out)
        tf = 3 * 2.0 * M * N * K / (us * 1e-6) / 1e12
        line.append(f"L{lay}: {us:6.1f} ({tf / 2516.8:.2f})")
    d = max((o - outs[0]).abs().max().item() / outs[0].abs().max().item() for o in outs)
    print(f"M={M:6d} N={N:5d} K={K:5d} ln={int(ln)} res={int(res)}  " + "  ".join(line) + f"  max|d| {d:.1e}",
          flush=True)
nn_ops.PANEL_FORM = 0


# ---- cold weights: every launch reads a DIFFERENT weight, 320 MB of images in rotation (more than the
# Infinity Cache holds), as the launches of a real step do; with and without the next-image hint
def bench_cold(M, N, K, ln, act, res, hint):
    g = torch.Generator().manual_seed(M + N + K)
    nw = max(8, int(320e6 / (N * K * 4)))
    xs = [torch.randn(M, K, generator=g).to(dev) for _ in range(8)]
    ws = [torch.nn.Parameter((torch.randn(N, K, generator=g) / K**0.5).to(dev), requires_grad=False) for _ in range(nw)]
    b = torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N, generator=g).to(dev) if res else None
    norms = None
    if ln:
        norms = [torch.nn.LayerNorm(K).to(dev) for _ in range(nw)]
        for n_ in norms:
            for p in n_.parameters():
                p.requires_grad_(False)
    nn_ops.PREFETCH_NEXT = hint
    nn_ops.prefetch_chain_reset()

    def run():
        for i in range(nw):
            nn_ops.linear(xs[i % 8], ws[i], b, r, act=act, alpha=0.5 if ln else 1.0, ln=norms[i] if ln else None)

    with torch.no_grad():
        run()
        run()   # (second pass: the launch order is known)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            graph.replay()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (2 * nw))
    return best, nw


if 3 in layouts and "--cold" in sys.argv:
    nn_ops.SPLIT_LAYOUT = 3
    print("cold weights (a different weight every launch, 320 MB in rotation), panel kernel: us per launch without / with the next-image hint")
    for shape in SHAPES[:4] + SHAPES[5:9]:
        M, N, K, ln, act, res = shape
        off, nw = bench_cold(*shape, hint=False)
        on, _ = bench_cold(*shape, hint=True)
        print(f"M={M:6d} N={N:5d} K={K:5d} ln={int(ln)}  {nw:4d} weights   no hint {off:7.1f} us   hint {on:7.1f} us", flush=True)
