#!/usr/bin/env python
"""Capture each stage of the joint step into a hipGraph separately and report which ones are
capturable / replay-identical.  Usage: python scripts/graph_debug.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from aps_amd.cplx import ComplexTensor  # noqa: E402


def try_capture(name, fn):
    try:
        for _ in range(2):
            ref = fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        g.replay()
        torch.cuda.synchronize()
        a = out[0] if isinstance(out, (tuple, list)) else out
        b = ref[0] if isinstance(ref, (tuple, list)) else ref
        a = a.real if isinstance(a, ComplexTensor) else a
        b = b.real if isinstance(b, ComplexTensor) else b
        print(f"[ok]   {name}: equal={torch.equal(a, b)}")
    except Exception as e:  # noqa: BLE001
        print(f"[FAIL] {name}: {str(e).splitlines()[0][:200]}")
        torch.cuda.synchronize()


with torch.no_grad():
    cpu, dev = bench.build_joint(torch.device("cuda", 0), 0)
    net, wav, lens = dev["net"], dev["wav"], dev["lens"]
    net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
    packed, n = net.enh_transform.encode(wav, lens)
    feats = net.enh_transform(packed)
    mask, _ = net.enh_net.mask_net(feats, n)
    ms, mn = torch.chunk(mask, 2, dim=-1)
    cst = ComplexTensor(packed[..., 0], packed[..., 1])
    y = net.enh_net.mvdr_net(ms, cst, x_len=n, mask_n=mn)
    x, _ = net.asr_transform(y, None)
    try_capture("encode", lambda: net.enh_transform.encode(wav, lens))
    try_capture("enh features", lambda: net.enh_transform(packed))
    try_capture("mask net", lambda: net.enh_net.mask_net(feats, n))
    try_capture("mvdr", lambda: net.enh_net.mvdr_net(ms, cst, x_len=n, mask_n=mn))
    try_capture("asr features", lambda: net.asr_transform(y, None))
    try_capture("conv2d proj", lambda: net.asr.encoder.proj(x, n))
    try_capture("asr", lambda: net.asr(x, n))
    try_capture("whole step", lambda: net(wav, lens))
