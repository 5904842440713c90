#!/usr/bin/env python
"""Timeline of ONE timed region of K = 20 steps of the headline pipeline (HIP events around every stage): when each
batch's stages begin and end relative to the region's start -- what the fill and the drain of the pipeline cost.
    python scripts/region_timeline_probe.py [K]"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from aps_amd.replicas import PipelinedReplicas  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, P = 6, 12
dev = torch.device("cuda:0")
_, d = bench.build_joint(dev, 0, P, 1)
net, wavs, lens = d["net"], d["wavs"], d["lens"]
net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
with torch.no_grad():
    for b in range(2):
        net(wavs[b], lens)
    torch.cuda.synchronize()
    reps = PipelinedReplicas([lambda b=b: net(wavs[b], lens) for b in range(P)], workers=W, lstm_share=2,
                             front="worker", mid="worker", lookahead=True)
for _ in range(2 * P):
    reps.submit(after_caller=False)
reps.synchronize()
for rep in range(2):
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    start.record()
    reps.stage_log = log = []
    for _ in range(K):
        reps.submit(after_caller=False)
    reps.flush()
    reps.synchronize()
    reps.stage_log = None
    # stages are logged in submission order: fronts (a, l) of step s at submit s, backs (m, b) at submit s + W / flush
    per = {"a": [], "l": [], "m": [], "b": []}
    for kind, e0, e1 in log:
        per[kind].append((start.elapsed_time(e0), start.elapsed_time(e1)))
    end = max(t1 for v in per.values() for _, t1 in v)
    print(f"region {rep}: {K} steps in {end:.2f} ms = {end / K:.3f} ms per step")
    for s in range(K):
        print(f"  step {s:2d}: " + "   ".join(f"{k.upper()} {per[k][s][0]:6.2f} -> {per[k][s][1]:6.2f}" for k in "almb"))
reps.close()
