#!/usr/bin/env python
"""Which stock torch kernels still run inside the joint step, and from which line of aps_amd?
One eager 32-utterance step under torch.profiler (with_stack): every device kernel that is not one of
the library's own (aps:: / aps_*) is listed with the aten op that launched it and the innermost
aps_amd frames of its Python stack.    python scripts/step_torch_ops.py [layers]
"""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
_, d = bench.build_joint(dev, 0, 1, 1)
net, wav, lens = d["net"], d["wavs"][0], d["lens"]
net.enh_transform.nan_policy = net.asr_transform.nan_policy = "deferred"
with torch.no_grad():
    for _ in range(3):
        net(wav, lens)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        net(wav, lens)
        torch.cuda.synchronize()

events = prof.events()
by_corr = {}
for e in events:
    if e.device_type == torch.autograd.DeviceType.CPU and e.stack:
        for k in e.kernels:
            by_corr[id(k)] = e
foreign = Counter()
for e in events:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    for k in e.kernels:
        name = k.name
        if name.startswith("aps") or "aps::" in name:
            continue
        frames = [f for f in (e.stack or []) if "aps_amd" in f or "bench.py" in f][:3]
        foreign[(name[:70], e.name, " <- ".join(f.strip()[-90:] for f in frames))] += 1
print(f"{sum(foreign.values())} launches of stock torch / runtime kernels in one step:")
for (kname, op, where), n in sorted(foreign.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {kname}\n      op {op}\n      at {where}")
