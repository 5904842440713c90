#!/usr/bin/env python
"""Recurrence launch alone (aps_lstm_layer on precomputed pre-activations; no GEMM, no host sync)
for DCCRN's pair geometry under APS_LSTM_SHAPE x APS_LSTM_DEBUG (1 = no gather, 2 = gather without
waiting for the sentinel, 3 = no gather and no publish).
   python scripts/lstm_shape_probe2.py [N T H]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aps_amd import _native as nat  # noqa: E402


def main():
    N, T, H = [int(v) for v in (sys.argv[1:4] + [64, 124, 512][len(sys.argv) - 1:])]
    lib = nat.load()
    torch.manual_seed(0)
    pre = [0.5 * torch.randn(N, T, 4 * H, device="cuda") for _ in range(2)]
    w = [torch.randn(4 * H, H, device="cuda") / H**0.5 for _ in range(2)]
    b = [0.1 * torch.randn(4 * H, device="cuda") for _ in range(2)]
    y = torch.empty(N, T, 2 * H, device="cuda")
    ws = torch.zeros(4, device="cuda", dtype=torch.int32)
    st = nat.stream_of(y)

    def call():
        rc = lib.aps_lstm_layer(nat.ptr(pre[0]), nat.ptr(pre[1]), nat.ptr(w[0]), nat.ptr(w[1]),
                                nat.ptr(b[0]), nat.ptr(b[1]), None, nat.ptr(y), N, T, H, 0, 1,
                                nat.ptr(ws), st)
        return rc

    for shp in ["auto", "4,1", "2,1", "2,2", "4,2", "1,2", "1,4", "2,4"]:
        if shp == "auto":
            os.environ.pop("APS_LSTM_SHAPE", None)
        else:
            os.environ["APS_LSTM_SHAPE"] = shp
        row = []
        for dbg in ["0", "1", "2", "3"]:
            os.environ["APS_LSTM_DEBUG"] = dbg
            rc = call()
            if rc != 0:
                row.append(f"rc={rc}")
                continue
            for _ in range(2):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            row.append(f"{e0.elapsed_time(e1) / 10 * 1e3 / T:6.2f}")
        print(f"N={N} T={T} H={H} shape {shp}: us/step  normal {row[0]} | no gather {row[1]} | "
              f"gather, no wait {row[2]} | no gather, no publish {row[3]}")


if __name__ == "__main__":
    main()
