#!/usr/bin/env python
"""Per-launch LSTM durations inside a replayed hipGraph (run under rocprofv3 --kernel-trace)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd import nn_ops  # noqa: E402

torch.manual_seed(0)
N, T, H = 32, 249, 512
with torch.no_grad():
    rnn = torch.nn.LSTM(512, H, 3, batch_first=True).eval().cuda()
    x = torch.randn(N, T, 512, device="cuda")
    lens = torch.full((N,), T, device="cuda", dtype=torch.int64) if "--lens" in sys.argv else None
    for _ in range(2):
        nn_ops.lstm_forward(rnn, x, lens)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = nn_ops.lstm_forward(rnn, x, lens)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    for _ in range(3):  # eager for comparison (later in the trace)
        nn_ops.lstm_forward(rnn, x, lens)
    torch.cuda.synchronize()
