#!/usr/bin/env python
"""Where do two-stream graph replays of the joint step differ from the eager step?
   python scripts/replica_diff.py [replicas] [group]

The regression run of the round-2 disturbance (DESIGN.md "co-residency"): with a library whose STFT is
built WITH packed-fp32 instructions and the 32-row bf16 GEMM as the other stream's kernel
(scripts/build_disturbance_libs.sh: APS_AMD_LIB=.../libaps_amd_dist_pkstft.so APS_GEMM_SPLIT_LAYOUT=1
APS_SPLIT_TM=32) it reports differing STFT stores every round; with the shipped flags (no packed-fp32
instructions anywhere) it reports none."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from aps_amd.replicas import GraphReplicas, _leaves  # noqa: E402

replicas = int(sys.argv[1]) if len(sys.argv) > 1 else 2
group = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
cpu, d = bench.build_joint(dev, 0, batches=3, group=group)
net, wavs, lens = d["net"], d["wavs"], d["lens"]
net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
with torch.no_grad(), warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for w in wavs:
        net(w, lens)
    torch.cuda.synchronize()
    a = [t.clone() for t in _leaves(net(wavs[0], lens))]
    b = [t.clone() for t in _leaves(net(wavs[0], lens))]
    torch.cuda.synchronize()
    print("eager twice equal:", all(torch.equal(x, y) for x, y in zip(a, b)))
    def step(k):
        # the front end stage by stage (what net.enhance does), then the whole step
        packed, xl = net.enh_transform.encode(wavs[k], lens)
        fe = net.enh_transform(packed)
        mn = net.enh_net.mask_net
        from aps_amd.nn_ops import linear
        h0 = linear(fe, mn.proj.weight, mn.proj.bias, act="relu")
        from aps_amd.asr.base.encoder import var_len_rnn_forward
        h1 = var_len_rnn_forward(mn.impl, h0, inp_len=xl, enforce_sorted=False)
        mask = linear(h1, mn.outp.weight, mn.outp.bias, act=mn.non_linear_name)
        feats, n = net.enhance(wavs[k], lens)
        return (packed, fe, h0, h1, mask, feats) + tuple(net(wavs[k], lens))

    reps = GraphReplicas([lambda k=k: step(k) for k in range(3)], replicas=replicas, verify=False)
    want = reps.eager_outputs if hasattr(reps, "eager_outputs") else None
    eager = [[t.clone() for t in _leaves(step(k))] for k in range(3)]
    torch.cuda.synchronize()
    bad_rounds = 0
    for rnd in range(int(os.environ.get("REPLICA_DIFF_ROUNDS", "12"))):
        for _ in range(3):
            reps.submit(after_caller=False)
        reps.synchronize()
        for k in range(3):
            for j, (x, y) in enumerate(zip(_leaves(reps.outputs[k]), eager[k])):
                if not torch.equal(x, y):
                    diff = (x.double() - y.double()).abs()
                    bad = (diff > 0)
                    idx = bad.nonzero()
                    if j > 1:
                        continue
                    print("   first differing indices:", idx[:6].tolist(), "last:", idx[-3:].tolist())
                    print(f"round {rnd} graph {k} tensor {j} shape {tuple(x.shape)}: {int(bad.sum())} elements differ, "
                          f"max {diff.max().item():.3e} (scale {y.abs().max().item():.3e}); utterances "
                          f"{sorted(set(idx[:, 0].tolist()))[:12]} frames {sorted(set(idx[:, 1].tolist()))[:12] if idx.shape[1] > 1 else ''}")
    from aps_amd import nn_ops
    print("lstm timeouts:", nn_ops.lstm_timeouts(dev))
    print("done")
