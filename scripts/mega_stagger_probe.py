#!/usr/bin/env python
"""aps_conformer_stack, 6 launches in flight: started together (all workgroups stream the same weights at the same time)
against started one sixth of a launch apart (what a pipeline does): per-launch duration from event pairs."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aps_amd import mega  # noqa: E402
from aps_amd.asr.transformer.impl import get_xfmr_encoder  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = get_xfmr_encoder("cfmr", "rel", 12, {"att_dim": 512, "nhead": 8, "feedforward_dim": 1024, "att_dropout": 0,
                                          "ffn_dropout": 0, "kernel_size": 15}).eval().to(dev)
N, T, R = 32, 63, 6
xs = [0.5 * torch.randn(N, T, 512, device=dev) for _ in range(R)]
rel = 0.1 * torch.randn(2 * T - 1, 64, device=dev)
mega.ENABLED = True
streams = [torch.cuda.Stream() for _ in range(R)]
with torch.no_grad():
    for i in range(R):
        enc.run(xs[i], None, rel=rel)
    torch.cuda.synchronize()
    graphs = []
    for i in range(R):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[i], capture_error_mode="thread_local"):
            enc.run(xs[i], None, rel=rel)
        graphs.append(g)

    def run(stagger_ms, rounds=6):
        marks = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(rounds):
            for i in range(R):
                with torch.cuda.stream(streams[i]):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(streams[i])
                    graphs[i].replay()
                    e1.record(streams[i])
                marks.append((e0, e1))
                if stagger_ms:
                    t_next = time.perf_counter() + stagger_ms * 1e-3
                    while time.perf_counter() < t_next:
                        pass
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        d = sorted(a.elapsed_time(b) for a, b in marks[R:])
        print(f"  stagger {stagger_ms:4.2f} ms: per-launch duration median {d[len(d) // 2]:.3f} ms (p10 {d[len(d) // 10]:.3f}, "
              f"p90 {d[9 * len(d) // 10]:.3f}); {1e3 * wall / (rounds * R):.3f} ms per batch")

    g1 = graphs[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g1.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"alone: {e0.elapsed_time(e1) / 4:.3f} ms")
    for s in (0.0, 0.3, 0.7, 1.0, 0.0):
        run(s)
