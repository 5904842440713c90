#!/usr/bin/env python
"""Do the two batches in flight fall into step (both in their LSTM phase, then both in their GEMMs)?
The joint step on two streams with the second stream's first replay delayed by a fraction of a step.
    python scripts/stagger_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from aps_amd.replicas import GraphReplicas  # noqa: E402

dev = torch.device("cuda:0")
cpu, d = bench.build_joint(dev, 0, batches=4, group=4)
net, wavs, lens = d["net"], d["wavs"], d["lens"]
net.enh_transform.nan_policy = net.asr_transform.nan_policy = "deferred"
with torch.no_grad():
    for w in wavs:
        net(w, lens)
    torch.cuda.synchronize()
    reps = GraphReplicas([(lambda w=w: net(w, lens)) for w in wavs], replicas=2, verify=False, guard_every=0)
    per_ms = bench.spin_cycles_for(1.0)
    for _ in range(8):
        reps.submit(after_caller=False)
    reps.synchronize()
    K = 60
    for offset_ms in (0.0, 1.0, 2.0, 3.5, 5.0, 0.0):
        best = None
        for _ in range(3):
            reps.synchronize()
            t0 = time.perf_counter()
            if offset_ms > 0:
                with torch.cuda.stream(reps.streams[1]):
                    torch.cuda._sleep(int(per_ms * offset_ms))
            for _ in range(K):
                reps.submit(after_caller=False)
            reps.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            best = ms if best is None else min(best, ms)
        print(f"second stream delayed by {offset_ms:3.1f} ms: {K} steps in {best:7.2f} ms = {best / K:.3f} ms per step "
              f"({(best - offset_ms) / K:.3f} without the delay itself)")
