#!/usr/bin/env python
"""What a low-latency form of the LAST encoder stages of a region would be worth (the drain of the pipeline: the last
batch's aps_conformer_stack launch runs alone on 32 CUs for 4 ms): regions of K steps where the last D steps run on a
second PipelinedReplicas captured with the per-launch encoder (the whole chip per projection), launched in the order a
unified pipeline would launch them.   python scripts/drain_form_probe.py [K]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from aps_amd import mega  # noqa: E402
from aps_amd.replicas import PipelinedReplicas  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, P = 6, 12
dev = torch.device("cuda:0")
_, d = bench.build_joint(dev, 0, P, 1)
net, wavs, lens = d["net"], d["wavs"], d["lens"]
net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
kw = dict(workers=W, lstm_share=2, front="worker", mid="worker", lookahead=True)
with torch.no_grad():
    for b in range(2):
        net(wavs[b], lens)
    torch.cuda.synchronize()
    reps = PipelinedReplicas([lambda b=b: net(wavs[b], lens) for b in range(P)], **kw)
    mega.ENABLED = False
    slow = PipelinedReplicas([lambda b=b: net(wavs[b], lens) for b in range(P)], **kw)
    mega.ENABLED = "auto"


def region(D):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K - D):
        reps.submit(after_caller=False)
    for s in range(D):                      # the last D steps: fronts of the per-launch pipeline, backs of the main one
        slow.submit(after_caller=False)
        if reps._pending:
            reps._finish_one()
    reps.flush()
    slow.flush()
    reps.synchronize()
    slow.synchronize()
    return 1e3 * (time.perf_counter() - t0)


for D in (0, 1, 2, 3, 0, 2, 3):
    for _ in range(2):
        region(D)
    ts = sorted(region(D) for _ in range(5))
    print(f"last {D} steps on the per-launch encoder: region of {K} steps median {ts[2]:.2f} ms = {ts[2] / K:.3f} ms per step "
          f"= {32 * K / ts[2] * 1e3:.0f} utt/s (min {ts[0]:.2f})", flush=True)
reps.close()
slow.close()
