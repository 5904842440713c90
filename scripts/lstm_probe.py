#!/usr/bin/env python
"""Time the persistent LSTM layer (N=32, T=249, H=512) under the kernel's timing probes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd import nn_ops  # noqa: E402

torch.manual_seed(0)
N, T, D, H = [int(v) for v in (sys.argv[1:5] + [32, 249, 512, 512][len(sys.argv) - 1:])]
with torch.no_grad():
    rnn = torch.nn.LSTM(D, H, 1, batch_first=True).eval().cuda()
    x = torch.randn(N, T, D, device="cuda")
    for dbg in ("0", "1", "2", "3"):
        os.environ["APS_LSTM_DEBUG"] = dbg
        for _ in range(3):
            nn_ops.lstm_forward(rnn, x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            nn_ops.lstm_forward(rnn, x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"debug={dbg}: {ms * 1e3:8.1f} us per layer call (incl. input GEMM), "
              f"{ms * 1e3 / T:6.2f} us / step")
