#!/usr/bin/env python
"""Why does aps_conformer_stack slow down with more launches in flight?  Cycles (s_memtime of workgroup 0 of ONE of the
launches) per layer and phase with R launches at once, next to the wall time per round: cycles that stay put while the
wall time grows = the clock came down; cycles that grow = the memory system.   APS_MEGA_TRACE=1 python scripts/mega_load_probe.py"""
import ctypes
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["APS_MEGA_TRACE"] = "1"
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aps_amd import _native, mega  # noqa: E402
from aps_amd.asr.transformer.impl import get_xfmr_encoder  # noqa: E402

L = 12
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = get_xfmr_encoder("cfmr", "rel", L, {"att_dim": 512, "nhead": 8, "feedforward_dim": 1024, "att_dropout": 0,
                                         "ffn_dropout": 0, "kernel_size": 15}).eval().to(dev)
N, T = 32, 63
xs = [0.5 * torch.randn(N, T, 512, device=dev) for _ in range(8)]
rel = 0.1 * torch.randn(2 * T - 1, 64, device=dev)
lib = _native.load()
buf = (ctypes.c_ulonglong * 32)()
streams = [torch.cuda.Stream() for _ in range(8)]
mega.ENABLED = True
names = ["ff1_up", "ff1_dn0", "ff1_dn1", "qkv", "attention", "out", "pw1", "glu_dwconv", "pw2", "ff2_up", "ff2_dn0", "ff2_dn1"]
with torch.no_grad():
    for i in range(8):
        enc.run(xs[i], None, rel=rel)
    torch.cuda.synchronize()
    for r in (1, 2, 4, 6, 8):
        lib.aps_debug_conformer_trace(buf)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            for i in range(r):
                with torch.cuda.stream(streams[i]):
                    enc.run(xs[i], None, rel=rel)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        lib.aps_debug_conformer_trace(buf)
        # (every launch's workgroup 0 adds to the same counters: r launches x reps)
        per = [buf[i] / (reps * r * L) for i in range(13)]
        tot = sum(per[:12])
        print(f"{r} in flight: {1e3 * dt:.3f} ms per round; workgroup 0: {tot:.0f} cycles per layer -> {tot * L / (dt * 1e9):.2f} GHz "
              f"if the workgroup ran the whole round | projections {sum(per[i] for i in (0,1,2,3,5,6,8,9,10,11)):.0f} "
              f"(staging {per[12]:.0f}) attention {per[4]:.0f} conv {per[7]:.0f}")
