#!/usr/bin/env python
"""LSTM recurrence time vs the placement of the exchange buffer y (base offset) and of pre."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd import _native as nat  # noqa: E402

lib = nat.load()
N, T, H = 32, 249, 512
torch.manual_seed(0)
with torch.no_grad():
    k = 1.0 / H**0.5
    w_hh = (torch.rand(4 * H, H, device="cuda") * 2 - 1) * k
    b_hh = (torch.rand(4 * H, device="cuda") * 2 - 1) * k
    big = torch.empty(N * T * H + (64 << 20) // 4, device="cuda")
    pbig = torch.randn(N * T * 4 * H + (64 << 20) // 4, device="cuda")
    ws = torch.empty(4, device="cuda", dtype=torch.int32)

    def run(yoff, poff, reps=5):
        y = big[yoff // 4: yoff // 4 + N * T * H].view(N, T, H)
        pre = pbig[poff // 4: poff // 4 + N * T * 4 * H].view(N, T, 4 * H)
        st = nat.stream_of(pre)
        for _ in range(2):
            assert lib.aps_lstm_layer(nat.ptr(pre), None, nat.ptr(w_hh), None, nat.ptr(b_hh), None,
                                      None, nat.ptr(y), N, T, H, nat.ptr(ws), st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            lib.aps_lstm_layer(nat.ptr(pre), None, nat.ptr(w_hh), None, nat.ptr(b_hh), None, None,
                               nat.ptr(y), N, T, H, nat.ptr(ws), st)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    print("y base", hex(big.data_ptr()), "pre base", hex(pbig.data_ptr()))
    for yoff in (0, 256, 1024, 4096, 16384, 65536, 1 << 20, 2 << 20, 3 << 20, 5 << 20, 8 << 20,
                 16 << 20, 32 << 20, (32 << 20) + 4096):
        print(f"y + {yoff:>9d}: {run(yoff, 0):7.1f} us")
    for poff in (4096, 1 << 20, 16 << 20):
        print(f"pre + {poff:>9d}: {run(0, poff):7.1f} us")
