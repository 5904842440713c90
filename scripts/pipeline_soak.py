#!/usr/bin/env python
"""Soak of the headline mode: PipelinedReplicas(workers=6, front / mid on the workers, lookahead) on the joint step of
BASELINE configs[4] for SECONDS of back-to-back submissions, every resident batch's outputs compared bit for bit with
the eager step every 600 submissions, hand-off time-outs and NaN rows counted at the end.
    python scripts/pipeline_soak.py [seconds]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from aps_amd import mega, nn_ops  # noqa: E402
from aps_amd.replicas import PipelinedReplicas  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
W, P = 6, 12
dev = torch.device("cuda:0")
_, d = bench.build_joint(dev, 0, P, 1)
net, wavs, lens = d["net"], d["wavs"], d["lens"]
net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
with torch.no_grad():
    for b in range(2):
        net(wavs[b], lens)
    torch.cuda.synchronize()
    calls0 = mega.CALLS
    reps = PipelinedReplicas([lambda b=b: net(wavs[b], lens) for b in range(P)], workers=W, lstm_share=2,
                             front="worker", mid="worker", lookahead=True)
assert mega.CALLS > calls0, "the conformer stack did not run as one launch per batch"
t0 = time.perf_counter()
n = checks = 0
while time.perf_counter() - t0 < SECONDS:
    for _ in range(600):
        reps.submit(after_caller=False)
    n += 600
    reps.synchronize()
    reps.check_outputs(reps.eager_outputs, f"after {n} submissions")
    checks += 1
dt = time.perf_counter() - t0
nans = net.enh_transform._nan_guard.count() + net.asr_transform._nan_guard.count()
print(f"{n} steps of 32 utterances in {dt:.1f} s ({32 * n / dt:.0f} utt/s incl. the checks' synchronisations), {checks} "
      f"bit-for-bit checks of all {P} resident batches passed, hand-off time-outs {nn_ops.lstm_timeouts(dev)}, "
      f"NaN rows {nans}, fp32-path blocks {nn_ops.fp16x2_wide_tiles(dev)}")
reps.close()
