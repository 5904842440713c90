#!/usr/bin/env python
"""Two streams, two kernels, no graphs: the 4-channel STFT of a fixed batch in a loop on one stream,
the bf16-split GEMM in a loop on the other; every STFT output is compared with the first one.
   python scripts/stft_vs_gemm_repro.py [iterations] [gemm: split|f32|none]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd import nn_ops  # noqa: E402
from aps_amd.transform import EnhTransform  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
kind = sys.argv[2] if len(sys.argv) > 2 else "split"
dev = torch.device("cuda:0")
torch.manual_seed(0)
tf = EnhTransform(feats="spectrogram-log-cmvn-ipd", frame_len=512, frame_hop=256, window="sqrthann",
                  ipd_index="0,1;0,2;0,3", cos_ipd=True).to(dev)
tf.nan_policy = "manual"
wav = 0.1 * torch.randn(128, 4, 64000, device=dev)
lens = torch.full((128,), 64000, device=dev)
M, N, K = 8064, 512, 512
x = torch.randn(M, K, device=dev)
w = torch.nn.Parameter(torch.randn(N, K, device=dev) / K**0.5, requires_grad=False)
w2 = torch.nn.Parameter(torch.randn(5000, K, device=dev) / K**0.5, requires_grad=False)
r = torch.randn(M, N, device=dev)
nn_ops.SPLIT_MODE = "1" if kind == "split" else "0"
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.no_grad():
    ref, _ = tf.encode(wav, lens)
    ref = ref.clone()
    gref = nn_ops.linear(x, w, None, residual=r).clone()
    torch.cuda.synchronize()
    bad = gbad = 0
    for it in range(iters):
        outs = []
        with torch.cuda.stream(s1):
            for _ in range(4):
                outs.append(tf.encode(wav, lens)[0])
        if kind != "none":
            with torch.cuda.stream(s2):
                for _ in range(6):
                    g = nn_ops.linear(x, w, None, residual=r)
                    nn_ops.linear(x, w2, None)
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                d = (o != ref)
                idx = d.nonzero()
                bad += 1
                if bad <= 5:
                    print(f"iteration {it}: STFT differs in {int(d.sum())} elements; first {idx[:3].tolist()}", flush=True)
        if kind != "none" and not torch.equal(g, gref):
            gbad += 1
    print(f"{kind}: {bad} corrupted STFT outputs of {4 * iters}, {gbad} corrupted GEMM outputs of {iters}")
