#!/usr/bin/env python
"""Round-2 sizing probe: (a) device time per GEMM launch at the conformer's shapes for 1 / 2 / 4
batches of 32 utterances per launch (M = 2016 / 4032 / 8064) and the mask estimator's shapes,
(b) the 2 x 512 LSTM stack for N = 32 / 64 at share 1 / 2.   python scripts/r02_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd import nn_ops  # noqa: E402
from aps_amd.nn_ops import linear  # noqa: E402
from aps_amd.replicas import concurrent_launches  # noqa: E402


def graph_time(fn, launches=20, replays=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(launches):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (launches * replays) * 1e3


def gemms():
    torch.manual_seed(0)
    for (N, K) in [(512, 512), (1024, 512), (512, 1024), (1536, 512)]:
        row = []
        for M in (2016, 4032, 8064):
            x = torch.randn(M, K, device="cuda")
            w = torch.randn(N, K, device="cuda") / K**0.5
            b = torch.randn(N, device="cuda")
            r = torch.randn(M, N, device="cuda")
            us = graph_time(lambda: linear(x, w, b, residual=r))
            row.append(f"M={M}: {us:6.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF")
        print(f"gemm N={N:5d} K={K:5d} | " + " | ".join(row), flush=True)
    for (M, N, K) in [(7968, 512, 1028), (7968, 2048, 512), (7968, 514, 512), (15936, 512, 1028),
                      (15936, 2048, 512), (15936, 514, 512), (2016, 5000, 512), (4032, 5000, 512)]:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K**0.5
        b = torch.randn(N, device="cuda")
        us = graph_time(lambda: linear(x, w, b))
        print(f"gemm {M} x {N} x {K}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF", flush=True)


def lstms():
    torch.manual_seed(1)
    rnn = torch.nn.LSTM(512, 512, num_layers=2, batch_first=True).eval().cuda()
    for N in (32, 64, 128):
        x = torch.randn(N, 249, 512, device="cuda")
        for share in (1, 2):
            with concurrent_launches(share):
                try:
                    us = graph_time(lambda: nn_ops.lstm_forward(rnn, x), launches=2, replays=5)
                    print(f"lstm 2x512 N={N} share={share}: {us / 1e3:6.3f} ms per call "
                          f"({us / N:6.1f} us / utterance), timeouts "
                          f"{nn_ops.lstm_timeouts(check=False)}", flush=True)
                except Exception as exc:  # noqa: BLE001
                    print(f"lstm N={N} share={share}: {exc}", flush=True)


if __name__ == "__main__":
    with torch.no_grad():
        gemms()
        lstms()
