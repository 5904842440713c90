#!/usr/bin/env python
"""Which projections of the joint FORWARD step (128 utterances) cost what: aps_linear_fp16x2 / aps_linear*
wrapped with HIP events per call behind a busy stream (launches back to back), grouped by (M, N, K).
   python scripts/joint_gemm_shapes.py [group]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from aps_amd import _native as nat  # noqa: E402

group = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
cpu, d = bench.build_joint(dev, 0, batches=1, group=group)
net, wav, lens = d["net"], d["wavs"][0], d["lens"]
net.enh_transform.nan_policy = net.asr_transform.nan_policy = "deferred"
lib = nat.load()
calls = []
SHAPE_ARGS = {"aps_linear_fp16x2": (9, 10, 11), "aps_linear": (6, 7, 8), "aps_linear_layernorm": (6, 7, 8)}
with torch.no_grad():
    for _ in range(2):
        net(wav, lens)
    for name, idx in SHAPE_ARGS.items():
        real = getattr(lib, name)

        def wrapped(*a, _real=real, _name=name, _idx=idx):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = _real(*a)
            e1.record()
            calls.append((_name, tuple(int(a[i]) for i in _idx), e0, e1))
            return rc
        setattr(lib, name, wrapped)
    per_ms = bench.spin_cycles_for(1.0)
    torch.cuda._sleep(int(per_ms * 30))  # the host runs ahead: launches sit back to back
    net(wav, lens)
    torch.cuda.synchronize()
by = collections.defaultdict(lambda: [0, 0.0])
for name, shape, e0, e1 in calls:
    by[(name, shape)][0] += 1
    by[(name, shape)][1] += e0.elapsed_time(e1) * 1e3
total = sum(v[1] for v in by.values())
print(f"{len(calls)} GEMM calls, {total / 1e3:.2f} ms between their event pairs (planes pass included, ~4.7 us of bracket each)")
for (name, shape), (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:16]:
    M, N, K = shape
    tf = 3 * 2.0 * M * N * K * n / (us * 1e-6) / 1e12 if "fp16x2" in name else 2.0 * M * N * K * n / (us * 1e-6) / 1e12
    peak = 2516.8 if "fp16x2" in name else 157.3
    print(f"  {name:22s} M,N,K={shape}  x{n:3d} {us:8.1f} us total {us / n:7.1f} us each  {tf:7.1f} TFLOP/s executed = {tf / peak:.2f} of its pipe")
