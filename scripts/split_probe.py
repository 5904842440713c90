#!/usr/bin/env python
"""Error of fp32 GEMMs evaluated as 3 / 6 bf16 MFMA products (scripts/micro/split_probe.hip)
against float64, beside the fp32 MFMA.   python scripts/split_probe.py"""
import ctypes
import os
import subprocess

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "micro", "split_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so,
                           os.path.join(here, "micro", "split_probe.hip")])
lib = ctypes.CDLL(so)
lib.split_probe.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
names = {0: "fp32 mfma", 1: "bf16 x3", 2: "bf16 x6", 3: "bf16 x6, hh apart", 4: "bf16 x6 small first",
         5: "fp16 x3"}
torch.manual_seed(0)
for kind in ("randn", "positive", "wide", "tiny rows", "mixed rows", "large"):
    for K in (512, 2048):
        M, N = 256, 256
        if kind == "randn":
            a, w = torch.randn(M, K), torch.randn(N, K) / K**0.5
        elif kind == "positive":  # no cancellation: relative error of the sum itself
            a, w = torch.rand(M, K) + 0.5, (torch.rand(N, K) + 0.5) / K
        elif kind == "tiny rows":  # every activation ~1e-4: the fp16 residuals are subnormal
            a, w = 1e-4 * torch.randn(M, K), torch.randn(N, K) / K**0.5
        elif kind == "mixed rows":  # rows of very different scale (judged per row below)
            a = torch.randn(M, K) * torch.logspace(-5, 3, M)[:, None]
            w = torch.randn(N, K) / K**0.5
        elif kind == "large":  # activations around 1e3 (fp16 max 65504)
            a, w = 3e3 * torch.randn(M, K), torch.randn(N, K) / K**0.5
        else:  # magnitudes over 12 orders
            a = torch.randn(M, K) * torch.exp(torch.randn(M, K) * 3)
            w = torch.randn(N, K) * torch.exp(torch.randn(N, K) * 3) / K**0.5
        ref = a.double() @ w.double().t()
        scale = ref.abs().max()
        ad, wd = a.cuda(), w.cuda()
        row = []
        for mode in range(6):
            c = torch.zeros(M, N, device="cuda")
            lib.split_probe(ad.data_ptr(), wd.data_ptr(), c.data_ptr(), M, N, K, mode, None)
            torch.cuda.synchronize()
            err = (c.cpu().double() - ref).abs()
            if kind == "mixed rows":  # worst row, each against its own scale
                per_row = (err.max(1).values / ref.abs().max(1).values).max()
                row.append(f"{names[mode]}: worst row {per_row:.2e}")
                continue
            row.append(f"{names[mode]}: max {err.max() / scale:.2e} rms {err.pow(2).mean().sqrt() / scale:.2e}")
        t = (ad @ wd.t()).cpu().double()
        row.append(f"torch fp32: max {(t - ref).abs().max() / scale:.2e}")
        print(f"[{kind} K={K}] " + " | ".join(row), flush=True)
