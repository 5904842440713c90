#!/usr/bin/env python
"""Which GEMM calls of the joint TRAINING step cost what: aps_linear wrapped with HIP events per call,
grouped by (M, N, K).   python scripts/train_gemm_shapes.py"""
import collections
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from aps_amd import _native as nat  # noqa: E402

dev = torch.device("cuda:0")
cpu, d = bench.build_joint(dev, 0, batches=1, group=1)
net, wav, lens = d["net"].train(), d["wavs"][0], d["lens"]
net.enh_transform.nan_policy = net.asr_transform.nan_policy = "deferred"
g = torch.Generator().manual_seed(11)
tgt = torch.randint(1, bench.JOINT_VOCAB, (bench.BATCH, 12), generator=g).to(dev)
tgt_len = torch.full((bench.BATCH,), 12, dtype=torch.int64, device=dev)


def step():
    net.zero_grad(set_to_none=True)
    _, enc_ctc, enc_len = net(wav, lens)
    logp = F.log_softmax(enc_ctc, -1).transpose(0, 1)
    F.ctc_loss(logp, tgt, enc_len, tgt_len, blank=0, reduction="mean", zero_infinity=True).backward()


for _ in range(2):
    step()
lib = nat.load()
calls = []
for name in ("aps_linear", "aps_linear_layernorm"):
    real = getattr(lib, name)

    def wrapped(*a, _real=real, _name=name):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = _real(*a)
        e1.record()
        ints = [int(v) for v in a if isinstance(v, int)]
        calls.append((_name, tuple(ints[:3]), e0, e1))
        return rc
    setattr(lib, name, wrapped)
step()
torch.cuda.synchronize()
by = collections.defaultdict(lambda: [0, 0.0])
for name, shape, e0, e1 in calls:
    by[(name, shape)][0] += 1
    by[(name, shape)][1] += e0.elapsed_time(e1) * 1e3
total = sum(v[1] for v in by.values())
print(f"{len(calls)} GEMM calls, {total / 1e3:.2f} ms between their event pairs (eager: includes launch gaps)")
for (name, shape), (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {name:22s} M,N,K={shape}  x{n:4d}  {us:9.1f} us total  {us / n:8.1f} us each")
