#!/usr/bin/env python
"""aps_conformer_stack alone and with R batches in flight: ms per 32-utterance batch of the 12-layer conformer stack of
BASELINE configs[4] (T = 63 frames), against the per-launch path (eager, one stream, captured as a graph).
    python scripts/mega_probe.py [layers]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aps_amd import mega, nn_ops  # noqa: E402
from aps_amd.asr.transformer.impl import get_xfmr_encoder  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = get_xfmr_encoder("cfmr", "rel", L, {"att_dim": 512, "nhead": 8, "feedforward_dim": 1024, "att_dropout": 0,
                                         "ffn_dropout": 0, "kernel_size": 15}).eval().to(dev)
N, T = 32, 63
R = 8
xs = [torch.randn(N, T, 512, device=dev) for _ in range(R)]
rel = 0.1 * torch.randn(2 * T - 1, 64, device=dev)
flops = L * 2 * N * T * (512 * 1024 * 4 + 512 * 1536 + 512 * 512 * 2 + 512 * 1024 + 0)   # projections only
with torch.no_grad():
    mega.ENABLED = False
    want = enc.run(xs[0], None, rel=rel)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out_g = enc.run(xs[0], None, rel=rel)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    per_launch_ms = 1e3 * (time.perf_counter() - t0) / 10
    print(f"per-launch path (hipGraph, one stream): {per_launch_ms:.3f} ms per batch of {N}")
    mega.ENABLED = True
    got = enc.run(xs[0], None, rel=rel)
    torch.cuda.synchronize()
    err = ((got - want).abs().max() / want.abs().max()).item()
    print(f"one launch per batch: max |diff| / scale = {err:.2e}")
    streams = [torch.cuda.Stream() for _ in range(R)]
    for r in (1, 2, 3, 4, 5, 6, 7, 8):
        for _ in range(2):
            for i in range(r):
                with torch.cuda.stream(streams[i]):
                    enc.run(xs[i], None, rel=rel)
        torch.cuda.synchronize()
        reps = 4
        t0 = time.perf_counter()
        for _ in range(reps):
            for i in range(r):
                with torch.cuda.stream(streams[i]):
                    enc.run(xs[i], None, rel=rel)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"  {r} in flight: {1e3 * dt:.3f} ms per round = {1e3 * dt / r:.3f} ms per batch, "
              f"{3 * flops * r / dt / 1e12:.0f} TFLOP/s executed ({3 * flops * r / dt / 2516.8e12:.3f} of the f16 peak)")
print("fp32-path blocks:", nn_ops.fp16x2_wide_tiles(dev))
if os.environ.get("APS_MEGA_TRACE") == "1":
    import ctypes
    from aps_amd import _native
    lib = _native.load()
    buf = (ctypes.c_ulonglong * 32)()
    lib.aps_debug_conformer_trace(buf)        # clear
    with torch.no_grad():
        for _ in range(4):
            enc.run(xs[0], None, rel=rel)
    torch.cuda.synchronize()
    assert lib.aps_debug_conformer_trace(buf) == 0
    names = ["ff1_up", "ff1_dn0", "ff1_dn1", "qkv", "attention", "out", "pw1", "glu_dwconv", "pw2", "ff2_up", "ff2_dn0",
             "ff2_dn1", "(staging, all projections)"]
    total = sum(buf[i] for i in range(12))
    print(f"workgroup 0, cycles per LAYER by phase (alone on the chip; {total / 4 / L:.0f} in all):")
    for i, nm in enumerate(names):
        print(f"   {nm:28s} {buf[i] / 4 / L:10.0f}")
