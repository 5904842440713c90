#!/usr/bin/env python
"""Where a workgroup of gemm_panel_kernel spends its life ALONE and UNDER LOAD: the s_memtime
stamps of the trace build (gemm_panel.hip, -DAPS_PANEL_TRACE: one slot per wave, handed out by an atomic, so launches
of several streams never share one) taken from the joint step of BASELINE configs[4] (32 utterances)
    alone      stage B of ONE batch replayed on one worker stream, nothing else on the chip
    3 workers  the headline mode: PipelinedReplicas(workers=3, lstm_share=2), head stream with the LSTM launches
    no LSTM    the same submissions with the head stream's stages left out (stage B's of three batches only)
segment by segment for the projections of one (N, K) at a time.

    scripts/build_variant_lib.sh ptrace gemm_panel gemm_panel.hip -DAPS_PANEL_TRACE
    APS_AMD_LIB=aps_amd/csrc/libaps_amd_ptrace.so python scripts/panel_trace_under_load.py
"""
import ctypes
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from aps_amd import _native  # noqa: E402
from aps_amd.replicas import PipelinedReplicas  # noqa: E402

SLOTS = 1 << 17
P = 6
dev = torch.device("cuda:0")
lib = _native.load()
lib.aps_debug_panel_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
lib.aps_debug_panel_trace.restype = ctypes.c_int64
lib.aps_debug_panel_trace_ctl.argtypes = [ctypes.c_int32] * 4 + [ctypes.c_void_p]
ctl_stream = torch.cuda.Stream()
buf = np.zeros(SLOTS * 32, dtype=np.uint64)


def ctl(n, k, record, reset):
    assert lib.aps_debug_panel_trace_ctl(n, k, record, reset, ctypes.c_void_p(ctl_stream.cuda_stream)) == 0


def read():
    used = lib.aps_debug_panel_trace(buf.ctypes.data, buf.nbytes)
    assert used >= 0
    return buf.reshape(SLOTS, 32)[:min(used, SLOTS)].astype(np.int64), used


_, d = bench.build_joint(dev, 0, P, 1)
net, wavs, lens = d["net"], d["wavs"], d["lens"]
net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
ctl(0, 0, 0, 1)
with torch.no_grad():
    for b in range(2):
        net(wavs[b], lens)
    torch.cuda.synchronize()
    reps = PipelinedReplicas([lambda b=b: net(wavs[b], lens) for b in range(P)], workers=3, lstm_share=2)
kinds = reps.kinds[0]
print("stages of a batch:", kinds)


def stage_b_only(i, stream):
    with torch.cuda.stream(stream):
        reps.pipelines[i][-1][0].replay()


def run(mode, rounds):
    """queue `rounds` passes over the resident batches in `mode`; recording is switched on behind the first pass"""
    def one(i):
        if mode == "alone":
            stage_b_only(0, reps.streams[0])
        elif mode == "no_lstm":
            stage_b_only(i % P, reps.streams[i % 3])
        else:
            reps.submit(after_caller=False)
    for i in range(P):
        one(i)
    ctl_now[0](1)
    for i in range(P, rounds * P):
        one(i)
    reps.synchronize()
    ctl_now[0](0)


ctl_now = [None]
SHAPES = [(512, 512), (1024, 512), (1536, 512), (512, 1024)]
for n, k in SHAPES:
    nch = (k + 127) // 128
    for mode, rounds in (("alone", 4), ("no_lstm", 4), ("pipeline", 4)):
        ctl(n, k, 0, 1)
        ctl_now[0] = lambda rec, n=n, k=k: ctl(n, k, rec, 0)
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        run(mode, rounds)
        t, used = read()
        if not len(t):
            print(f"N={n} K={k} {mode}: nothing recorded")
            continue
        ok = (t[:, 27] > t[:, 0]) & (t[:, 0] > 0)
        t = t[ok]
        print(f"N={n} K={k} ({nch} chunks of 128) {mode}: {used} wave slots handed out, {len(t)} complete; "
              f"wave life mean {np.mean(t[:, 27] - t[:, 0]):.0f} cycles (p10 {np.percentile(t[:, 27] - t[:, 0], 10):.0f}, "
              f"p90 {np.percentile(t[:, 27] - t[:, 0], 90):.0f}); LN launches {int(((t[:, 31] >> 8) & 1).sum())}")
        seg = [("entry -> first rows arrived", 0, 1)]
        split = mfma = bar = 0
        for c in range(nch):
            first = 1 if c == 0 else 4 + 3 * (c - 1)
            split = split + (t[:, 2 + 3 * c] - t[:, first])
            bar = bar + (t[:, 3 + 3 * c] - t[:, 2 + 3 * c])
            mfma = mfma + (t[:, 4 + 3 * c] - t[:, 3 + 3 * c])
        d01 = t[:, 1] - t[:, 0]
        tail = t[:, 26] - t[:, 4 + 3 * (nch - 1)]
        epi = t[:, 27] - t[:, 26]
        for name, v in (("entry -> first rows arrived", d01), (f"{nch} x maxima + split + LDS writes", split),
                        (f"{nch} x barrier A", bar), (f"{nch} x MFMA loop + fold + barrier B", mfma),
                        ("last barrier -> epilogue (wide check, stats)", tail), ("epilogue (+ prefetch, stores drained)", epi)):
            print(f"    {name:48s} mean {v.mean():8.0f}   p10 {np.percentile(v, 10):8.0f}   p90 {np.percentile(v, 90):8.0f}")
        first_mfma = t[:, 4] - t[:, 3]
        print(f"    (first chunk's MFMA loop alone: mean {first_mfma.mean():.0f}; a wave's MFMAs occupy its SIMD for "
              f"{(k // 32) * 6 * 32} cycles in all)")
        xcc = (t[:, 30] >> 32) & 0xf
        print("    waves per XCC:", np.bincount(xcc, minlength=8).tolist())
reps.close()
