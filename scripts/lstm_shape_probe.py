#!/usr/bin/env python
"""Persistent LSTM decompositions (MT row tiles x UT unit tiles per workgroup, APS_LSTM_SHAPE) on
the two benchmark geometries, timed on one box and checked against torch's library LSTM:
  pair : DCCRN's real / imaginary pair, N = 64, T = 124, D = 768, H = 512 (one layer, two LSTMs)
  stack: the mask estimator's 2-layer stack, N = 32, T = 249, H = 512 (one pipelined launch)
   python scripts/lstm_shape_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd import nn_ops  # noqa: E402

SHAPES = ["", "4,1", "2,1", "1,1", "2,2", "1,2", "1,4"]


def timed(fn, reps=10):
    for _ in range(3):
        out = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, out


def main():
    torch.manual_seed(0)
    nn_ops.LSTM_CHECK = True
    with torch.no_grad():
        # ---- pair (one layer): time = 2 input GEMMs + the recurrence launch
        a = torch.nn.LSTM(768, 512, 1, batch_first=True).eval().cuda()
        b = torch.nn.LSTM(768, 512, 1, batch_first=True).eval().cuda()
        x = torch.randn(64, 124, 768, device="cuda")
        ref_a, ref_b = a(x)[0], b(x)[0]
        for shp in SHAPES:
            if shp:
                os.environ["APS_LSTM_SHAPE"] = shp
            else:
                os.environ.pop("APS_LSTM_SHAPE", None)
            try:
                us, (ya, yb) = timed(lambda: nn_ops.lstm_pair_forward(a, b, x))
            except RuntimeError as exc:
                print(f"pair  shape {shp or 'auto':5s}: {str(exc)[:90]}")
                continue
            err = max((ya - ref_a).abs().max().item(), (yb - ref_b).abs().max().item())
            print(f"pair  shape {shp or 'auto':5s}: {us:8.1f} us / call, {us / 124:6.2f} us / step, "
                  f"max err {err:.1e}")
        # ---- stack
        rnn = torch.nn.LSTM(512, 512, 2, batch_first=True).eval().cuda()
        x = torch.randn(32, 249, 512, device="cuda")
        lens = torch.tensor([249] * 20 + [200] * 12, device="cuda")
        ref = rnn(x)[0]
        for shp in SHAPES:
            if shp:
                os.environ["APS_LSTM_SHAPE"] = shp
            else:
                os.environ.pop("APS_LSTM_SHAPE", None)
            try:
                us, y = timed(lambda: nn_ops.lstm_forward(rnn, x))
                _, yl = timed(lambda: nn_ops.lstm_forward(rnn, x, lens), reps=1)
            except RuntimeError as exc:
                print(f"stack shape {shp or 'auto':5s}: {str(exc)[:90]}")
                continue
            err = (y - ref).abs().max().item()
            errl = max((yl[:20] - ref[:20]).abs().max().item(),
                       (yl[20:, :200] - ref[20:, :200]).abs().max().item(),
                       yl[20:, 200:].abs().max().item())
            print(f"stack shape {shp or 'auto':5s}: {us:8.1f} us / call, {us / 249:6.2f} us / step, "
                  f"max err {err:.1e} (ragged {errl:.1e})")
        os.environ.pop("APS_LSTM_SHAPE", None)
        os.environ["APS_NO_LSTM_STACK"] = "1"


if __name__ == "__main__":
    main()
