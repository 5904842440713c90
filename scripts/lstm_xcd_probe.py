#!/usr/bin/env python
"""The LSTM hand-off when every workgroup of a launch sits on one XCD (experiment flags of
APS_LSTM_DEBUG: 4 = only blocks with id % 8 == 0 take part, 8 = plain stores instead of sc1).
    python scripts/lstm_xcd_probe.py [N T D H] ; APS_LSTM_SHAPE=MT,UT picks the decomposition"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd import nn_ops  # noqa: E402

torch.manual_seed(0)
N, T, D, H = [int(v) for v in (sys.argv[1:5] + [16, 249, 512, 512][len(sys.argv) - 1:])]
with torch.no_grad():
    rnn = torch.nn.LSTM(D, H, 1, batch_first=True).eval().cuda()
    x = torch.randn(N, T, D, device="cuda")
    ref = None
    for dbg, what in (("0", "spread over the XCDs, sc1 stores (shipped)"), ("4", "one XCD, sc1 stores"),
                      ("12", "one XCD, plain stores"), ("8", "spread, plain stores (NOT a valid protocol)"),
                      ("3", "no gather, no publish")):
        os.environ["APS_LSTM_DEBUG"] = dbg
        nn_ops.LSTM_CHECK = True
        try:
            for _ in range(3):
                out = nn_ops.lstm_forward(rnn, x)
            err = "timeouts: none"
        except RuntimeError as e:
            err = "TIMED OUT: " + str(e)[:60]
        nn_ops.LSTM_CHECK = False
        if ref is None:
            ref = out.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            out = nn_ops.lstm_forward(rnn, x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        same = "identical" if torch.equal(out, ref) else f"max diff {(out - ref).abs().max().item():.2e}"
        print(f"debug={dbg:>2s} {what:45s}: {ms * 1e3 / T:6.2f} us / step   output {same}; {err}")
