#!/usr/bin/env python
"""aps_linear_split against aps_linear at the merged-batch conformer shapes: error vs float64 and
device time per launch (20 launches in a captured graph).   python scripts/split_gemm_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd import nn_ops  # noqa: E402
from scripts.r02_probe import graph_time  # noqa: E402

torch.manual_seed(0)
shapes = [(512, 512), (1024, 512), (1536, 512), (2048, 512), (512, 2048), (514, 1028)]
Ms = [int(m) for m in sys.argv[1:]] or [2016, 8064]
with torch.no_grad():
    for M in Ms:
        for (N, K) in shapes:
            x = torch.randn(M, K, device="cuda")
            w = torch.nn.Parameter(torch.randn(N, K, device="cuda") / K**0.5, requires_grad=False)
            b = torch.randn(N, device="cuda")
            r = torch.randn(M, N, device="cuda")
            ln = torch.nn.LayerNorm(K).cuda()
            ln.weight.data.uniform_(0.5, 1.5)
            ln.bias.data.normal_(0, 0.2)
            ref = (x.double() @ w.double().t() + b.double()) * 0.5 + r.double()
            ref_ln = torch.nn.functional.silu(
                torch.nn.functional.layer_norm(x.double(), (K,), ln.weight.double(), ln.bias.double())
                @ w.double().t() + b.double())
            row = []
            variants = [("fp32", "0", None, None), ("bd", "1", None, "bd"), ("fp16", "1", None, "fp16"),
                        ("pc", "1", "128", "pc")]
            for swp in ("0", "1"):
                for tn in ("64", "128"):
                    variants.append((("swp" if swp == "1" else "v1") + tn, "1", tn, swp))
            if os.environ.get("SPLIT_BENCH_ONLY"):
                variants = [v for v in variants if v[0] in os.environ["SPLIT_BENCH_ONLY"].split(",")]
            for tag, mode, tn, swp in variants:
                nn_ops.SPLIT_MODE = mode
                nn_ops.SPLIT_LAYOUT = {"bd": 1, "fp16": 2}.get(swp, 0)
                if tn:
                    os.environ["APS_SPLIT_TN"] = tn
                    os.environ["APS_SPLIT_KERNEL"] = "pc" if swp == "pc" else ("swp" if swp == "1" else "v1")
                got = nn_ops.linear(x, w, b, residual=r, alpha=0.5)
                err = ((got.double() - ref).abs().max() / ref.abs().max()).item()
                got_ln = nn_ops.linear(x, w, b, act="swish", ln=ln)
                err_ln = ((got_ln.double() - ref_ln).abs().max() / ref_ln.abs().max()).item()
                us = graph_time(lambda: nn_ops.linear(x, w, b, residual=r, alpha=0.5))
                us_ln = graph_time(lambda: nn_ops.linear(x, w, b, act="swish", ln=ln))
                row.append(f"{tag}: {us:6.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF err {err:.1e}"
                           f" | ln {us_ln:6.1f} us err {err_ln:.1e}")
            os.environ.pop("APS_SPLIT_TN", None)
            os.environ.pop("APS_SPLIT_KERNEL", None)
            print(f"M={M} N={N} K={K} || " + " || ".join(row), flush=True)
