#!/usr/bin/env python
"""aps_conformer_stack with R launches in flight, each stream confined to ONE XCD by a CU mask
(hipExtStreamCreateWithCUMask): a launch is 32 workgroups of a whole CU each and an XCD has 32 CUs, so a pinned
launch reads its weights through ONE L2 (fetched from the fabric once per launch instead of once per XCD = 8 x) and
its 32 workgroups stream the same fragments at about the same time.
    python scripts/mega_cumask_probe.py [layout: interleaved | contiguous | none]"""
import ctypes
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aps_amd import mega, nn_ops  # noqa: E402
from aps_amd.asr.transformer.impl import get_xfmr_encoder  # noqa: E402

layout = sys.argv[1] if len(sys.argv) > 1 else "interleaved"
L = 12
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = get_xfmr_encoder("cfmr", "rel", L, {"att_dim": 512, "nhead": 8, "feedforward_dim": 1024, "att_dropout": 0,
                                         "ffn_dropout": 0, "kernel_size": 15}).eval().to(dev)
N, T, R = 32, 63, 8
xs = [torch.randn(N, T, 512, device=dev) for _ in range(R)]
rel = 0.1 * torch.randn(2 * T - 1, 64, device=dev)
flops = L * 2 * N * T * (512 * 1024 * 4 + 512 * 1536 + 512 * 512 * 2 + 512 * 1024)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]


def masked_stream(xcd: int):
    words = (ctypes.c_uint32 * 8)()
    for cu in range(256):
        on = (cu % 8 == xcd) if layout == "interleaved" else (cu // 32 == xcd)
        if on:
            words[cu // 32] |= 1 << (cu % 32)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(h.value, device=dev)


torch.cuda.init()
torch.zeros(1, device=dev)
streams = [torch.cuda.Stream() for _ in range(R)] if layout == "none" else [masked_stream(i) for i in range(R)]
mega.ENABLED = True
with torch.no_grad():
    want = enc.run(xs[0], None, rel=rel).clone()
    torch.cuda.synchronize()
    with torch.cuda.stream(streams[0]):
        got = enc.run(xs[0], None, rel=rel)
    torch.cuda.synchronize()
    print(f"layout {layout}: masked-stream result equals the default stream's: {torch.equal(got, want)}")
    for r in (1, 2, 4, 6, 8):
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
              f"{3 * flops * r / dt / 2516.8e12:.3f} of the f16 peak", flush=True)
    # a graph captured on a masked stream and replayed there (what PipelinedReplicas would do)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(streams[0]):
        with torch.cuda.graph(g, stream=streams[0]):
            out_g = enc.run(xs[0], None, rel=rel)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
    print(f"  graph replay on stream 0: {1e2 * (time.perf_counter() - t0):.3f} ms per launch; equal: {torch.equal(out_g, want)}")
print("fp32-path blocks:", nn_ops.fp16x2_wide_tiles(dev))
# is the mask honoured?  64 workgroups on a stream confined to 32 CUs take two rounds
x64 = torch.randn(64, T, 512, device=dev)
with torch.no_grad():
    for st, nm in ((streams[0], "probe stream 0"), (torch.cuda.Stream(), "an unmasked stream")):
        with torch.cuda.stream(st):
            for _ in range(2):
                enc.run(x64, None, rel=rel)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                enc.run(x64, None, rel=rel)
            torch.cuda.synchronize()
        print(f"  64 utterances in one launch on {nm}: {250 * (time.perf_counter() - t0):.3f} ms")
if os.environ.get("APS_MEGA_TRACE") == "1":
    from aps_amd import _native
    lib = _native.load()
    buf = (ctypes.c_ulonglong * 32)()
    names = ["ff1_up", "ff1_dn0", "ff1_dn1", "qkv", "attention", "out", "pw1", "glu_dwconv", "pw2", "ff2_up", "ff2_dn0",
             "ff2_dn1", "(staging, all projections)"]
    for r in (1, 8):
        lib.aps_debug_conformer_trace(buf)        # clear
        reps = 4
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(reps):
                for i in range(r):
                    with torch.cuda.stream(streams[i]):
                        enc.run(xs[i], None, rel=rel)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        assert lib.aps_debug_conformer_trace(buf) == 0
        total = sum(buf[i] for i in range(12)) / (reps * r)
        print(f"{r} in flight: workgroup 0 of every launch, s_memtime ticks per LAUNCH {total:.0f} over {1e3 * wall:.3f} ms per round "
              f"-> {total / wall / 1e9:.3f} G ticks / s; per layer by phase:")
        print("   " + "  ".join(f"{nm} {buf[i] / (reps * r) / L:.0f}" for i, nm in enumerate(names)))
