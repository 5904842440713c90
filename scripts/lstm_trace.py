#!/usr/bin/env python
"""Where a step of lstm_layer_kernel spends its cycles: s_memtime stamps of lane 0 of every wave of
two workgroups, summed over the steps (a library built with -DAPS_LSTM_TRACE, selected through
APS_AMD_LIB).

    scripts/build_variant_lib.sh lstmtrace lstm lstm.hip -DAPS_LSTM_TRACE
    APS_AMD_LIB=aps_amd/csrc/libaps_amd_lstmtrace.so python scripts/lstm_trace.py [N T D H]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aps_amd import _native, nn_ops  # noqa: E402

N, T, D, H = [int(v) for v in (sys.argv[1:5] + [128, 249, 512, 512][len(sys.argv) - 1:])]
lib = _native.load()
lib.aps_debug_lstm_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
names = ["wait for the gather (finish)", "split + LDS writes", "barrier 1", "issue next gather",
         "LDS reads + MFMAs + partial sums to LDS", "barrier 2", "gate threads: partial sums",
         "gate math", "shuffles + publish"]
torch.manual_seed(0)
with torch.no_grad():
    rnn = torch.nn.LSTM(D, H, 1, batch_first=True).eval().cuda()
    x = torch.randn(N, T, D, device="cuda")
    for dbg in ("0", "3"):
        os.environ["APS_LSTM_DEBUG"] = dbg
        for _ in range(3):
            nn_ops.lstm_forward(rnn, x)
        torch.cuda.synchronize()
        buf = np.zeros(2 * 2 * 4 * 16, dtype=np.uint64)
        assert lib.aps_debug_lstm_trace(buf.ctypes.data, buf.nbytes) == 0
        t = buf.reshape(2, 2, 4, 16).astype(np.float64)
        steps = t[0, 0, 0, 15]
        print(f"== APS_LSTM_DEBUG={dbg} ({'no gather, no publish' if dbg == '3' else 'the real hand-off'}): "
              f"N={N} T={T} H={H}, cycles per step (mean over {int(steps)} steps)")
        for slot in range(2):
            for g in range(2):
                seg = t[slot, g, :, :9] / steps
                if seg.sum() == 0:
                    continue
                print(f"  workgroup slot {slot}, row group {g}: {seg.sum(1).mean():7.0f} cycles per step")
                for q, nm in enumerate(names):
                    print(f"      {nm:42s} {seg[:, q].mean():7.0f}   waves {np.round(seg[:, q]).astype(int).tolist()}")
