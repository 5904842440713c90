#!/usr/bin/env python
"""Why is the 2nd LSTM launch of the joint graph faster (0.77 vs 1.08 ms)?  Capture the mask net
alone in variants; run under rocprofv3 --kernel-trace and read the per-launch durations."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd.asr.base.encoder import PyTorchRNNEncoder  # noqa: E402

torch.manual_seed(0)
N, T = 32, 249
variant = sys.argv[1] if len(sys.argv) > 1 else "full"
with torch.no_grad():
    net = PyTorchRNNEncoder(1028, 514, input_proj=512, rnn="lstm", num_layers=2, hidden=512,
                            dropout=0.0, bidirectional=False, non_linear="sigmoid").eval().cuda()
    feats = torch.randn(N, T, 1028, device="cuda")
    lens = torch.full((N,), T, device="cuda", dtype=torch.int64)
    if variant == "nolens":
        lens = None
    if variant == "noproj":
        net.proj = None
        net.impl = torch.nn.LSTM(1028, 512, 2, batch_first=True).cuda()
    for _ in range(2):
        net(feats, lens)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = net(feats, lens)
    for _ in range(6):
        g.replay()
    torch.cuda.synchronize()
