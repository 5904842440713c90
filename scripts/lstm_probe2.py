#!/usr/bin/env python
"""Is the LSTM recurrence time data dependent?  Times aps_lstm_layer alone for differently scaled
pre-activations / weights (N=32, T=249, H=512)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd import _native as nat  # noqa: E402

lib = nat.load()
N, T, H = 32, 249, 512
torch.manual_seed(0)


def run(pre, w_hh, b_hh, reps=10):
    y = torch.empty(N, T, H, device="cuda")
    ws = torch.empty(4, device="cuda", dtype=torch.int32)
    st = nat.stream_of(pre)

    def once():
        rc = lib.aps_lstm_layer(nat.ptr(pre), None, nat.ptr(w_hh), None, nat.ptr(b_hh), None, None,
                                nat.ptr(y), N, T, H, nat.ptr(ws), st)
        assert rc == 0
    for _ in range(3):
        once()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    assert lib.aps_lstm_timed_out(nat.ptr(ws), st) == 0
    return e0.elapsed_time(e1) / reps * 1e3, y


with torch.no_grad():
    k = 1.0 / H**0.5
    w_hh = (torch.rand(4 * H, H, device="cuda") * 2 - 1) * k
    b_hh = (torch.rand(4 * H, device="cuda") * 2 - 1) * k
    for scale in (0.0, 0.1, 1.0, 10.0, 100.0):
        pre = torch.randn(N, T, 4 * H, device="cuda") * scale
        us, y = run(pre, w_hh, b_hh)
        print(f"pre ~ N(0, {scale}^2): {us:8.1f} us = {us / T:5.2f} us/step   |h| mean {y.abs().mean().item():.3f}")
    pre = torch.randn(N, T, 4 * H, device="cuda")
    for ws_ in (0.0, 1.0, 10.0):
        us, y = run(pre, w_hh * ws_, b_hh)
        print(f"w_hh x {ws_}: {us:8.1f} us = {us / T:5.2f} us/step")

# ---- does what ran before matter?  (in the joint step layer 2 takes 0.77 ms, layer 1 1.08 ms)
from aps_amd.nn_ops import linear  # noqa: E402

with torch.no_grad():
    pre = torch.randn(N, T, 4 * H, device="cuda")
    x = torch.randn(N * T, 512, device="cuda")
    w = torch.randn(2048, 512, device="cuda")
    ys = [torch.empty(N, T, H, device="cuda") for _ in range(4)]
    ws = torch.empty(4, device="cuda", dtype=torch.int32)
    st = nat.stream_of(pre)

    def lstm(y, p):
        assert lib.aps_lstm_layer(nat.ptr(p), None, nat.ptr(w_hh), None, nat.ptr(b_hh), None, None,
                                  nat.ptr(y), N, T, H, nat.ptr(ws), st) == 0

    def timed(fn, reps=6):
        evs = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn(a, b)
            evs.append((a, b))
        torch.cuda.synchronize()
        return [round(a.elapsed_time(b) * 1e3) for a, b in evs]

    def alone(a, b):
        a.record(); lstm(ys[0], pre); b.record()

    def after_gemm(a, b):
        p2 = linear(x, w).view(N, T, 4 * H)
        a.record(); lstm(ys[1], p2); b.record()

    def chained(a, b):  # layer 1 -> GEMM on its output -> layer 2 (timed)
        lstm(ys[2], pre)
        p2 = linear(ys[2].view(N * T, H), w).view(N, T, 4 * H)
        a.record(); lstm(ys[3], p2); b.record()

    print("alone            :", timed(alone))
    print("after a GEMM     :", timed(after_gemm))
    print("layer-2 position :", timed(chained))
