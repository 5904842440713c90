#!/usr/bin/env python
"""One chained launch (aps_linear_chain) of a conformer layer's first three projections at M rows, against the
same three launches one by one: bit-equality, the time of both forms, the chain's error word.
    python scripts/chain_smoke.py [M] [workgroups]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from aps_amd import nn_ops  # noqa: E402
from test_gpu_encoder import _chain_case  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2016
nn_ops.CHAIN_WORKGROUPS = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
with torch.no_grad():
    x, stages = _chain_case(M, dev, 1)
    nn_ops.CHAIN = False
    want = nn_ops.linear_chain(x, stages)
    torch.cuda.synchronize()
    nn_ops.CHAIN = True
    t0 = time.perf_counter()
    got = nn_ops.linear_chain(x, stages)
    torch.cuda.synchronize()
    print(f"first chained launch: {1e3 * (time.perf_counter() - t0):.2f} ms (host clock, incl. setup)", flush=True)
    for k, (a, b) in enumerate(zip(got, want)):
        print(f"stage {k}: equal {torch.equal(a, b)}  differing {int((a != b).sum())} of {a.numel()}", flush=True)
    print("expired waits:", nn_ops.chain_errors(dev), flush=True)
    ws = nn_ops.chain_workspace(dev, create=False)
    print("workspace sum after the launch:", int(ws.abs().sum().item()), flush=True)
    for chain in (False, True):
        nn_ops.CHAIN = chain
        for _ in range(3):
            nn_ops.linear_chain(x, stages)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            nn_ops.linear_chain(x, stages)
        e1.record()
        torch.cuda.synchronize()
        print(f"chain={chain}: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us per 3 projections (eager, back to back)", flush=True)
    print("expired waits:", nn_ops.chain_errors(dev), flush=True)
