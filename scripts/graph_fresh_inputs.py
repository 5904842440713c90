#!/usr/bin/env python
"""Replay the captured joint step on CHANGING inputs and compare with eager execution: guards the
LSTM hand-off (sentinel protocol) against consuming a previous replay's data."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

with torch.no_grad():
    cpu, dev = bench.build_joint(torch.device("cuda", 0), 0)
    net, lens = dev["net"], dev["lens"]
    net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
    static = dev["wav"].clone()
    for _ in range(2):
        net(static, lens)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = net(static, lens)
    worst = 0.0
    for k in range(4):
        gen = torch.Generator().manual_seed(100 + k)
        x = (0.1 * torch.randn(static.shape, generator=gen)).cuda() * (1 + k)
        static.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        got = out[0].clone()
        ref = net(x, lens)[0]
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        worst = max(worst, err)
        print(f"replay {k}: max scaled difference graph vs eager {err:.3e}")
    assert worst == 0.0, "graph replay on fresh inputs differs from eager execution"
    print("OK")
