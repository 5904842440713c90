#!/usr/bin/env python
"""Device time per GEMM launch (20 launches in a captured graph) at the shapes of the joint step and
the encoder workload, plain and with the folded LayerNorm, with correctness against torch.
APS_AMD_LIB=<other build> runs the same table on another library of the same ABI (A/B on one box).
   python scripts/gemm_shape_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd.nn_ops import linear  # noqa: E402

SHAPES = [(2016, 512, 512), (2016, 1024, 512), (2016, 512, 1024), (2016, 1536, 512),
          (7968, 2048, 512), (7968, 512, 1028), (2016, 5000, 512), (12800, 2048, 512),
          (12800, 512, 2048), (12800, 1536, 512), (4096, 4096, 4096)]

with torch.no_grad():
    tag = os.path.basename(os.environ.get("APS_AMD_LIB", "libaps_amd.so"))
    torch.manual_seed(0)
    for M, N, K in SHAPES:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K**0.5
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda")
        ln = torch.nn.LayerNorm(K).cuda()
        ln.weight.data.uniform_(0.5, 1.5)
        ln.bias.data.normal_()
        y = linear(x, w, b, residual=r, act="swish")
        ref = torch.nn.functional.silu(x.double() @ w.double().t() + b.double()) + r.double()
        err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
        yl = linear(x, w, b, ln=ln)
        refl = ln(x).double() @ w.double().t() + b.double()
        errl = ((yl.double() - refl).abs().max() / refl.abs().max()).item()
        res = []
        for use_ln in (False, True):
            outs = [torch.empty(M, N, device="cuda") for _ in range(2)]
            g = torch.cuda.CUDAGraph()
            linear(x, w, b, residual=r, ln=ln if use_ln else None)
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                for _ in range(20):
                    linear(x, w, b, residual=r, ln=ln if use_ln else None)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 100 * 1e3
            res.append(f"{us:7.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF")
        print(f"{tag} {M:6d} x {N:5d} x {K:5d}: plain {res[0]} | LN-fused {res[1]} | err {err:.1e} / {errl:.1e}")
