#!/usr/bin/env python
"""Where a workgroup of gemm_panel_kernel spends its life ALONE and UNDER LOAD: the s_memtime
stamps of the trace build (gemm_panel.hip, -DAPS_PANEL_TRACE: one slot per workgroup, handed out by the HOST when a
launch is issued / captured -- no atomic in the kernel -- stamped by lane 0 of wave 0) taken from the joint step of
BASELINE configs[4] (32 utterances)
    alone      stage B of ONE batch replayed on one worker stream, nothing else on the chip
    no_lstm    stage B of three batches on the three worker streams, the head stream idle
    pipeline   the headline mode: PipelinedReplicas(workers=3, lstm_share=2), head stream with A / LSTM / M stages
per (N, K) of the conformer's projections, segment by segment.

    scripts/build_variant_lib.sh ptrace gemm_panel gemm_panel.hip -DAPS_PANEL_TRACE
    APS_AMD_LIB=aps_amd/csrc/libaps_amd_ptrace.so [APS_PANEL_FORM=f] python scripts/panel_trace_under_load.py
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

SLOTS = 1 << 19
P = 3
dev = torch.device("cuda:0")
lib = _native.load()
lib.aps_debug_panel_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
lib.aps_debug_panel_trace.restype = ctypes.c_int64
lib.aps_debug_panel_trace_filter.argtypes = [ctypes.c_int32] * 3
lib.aps_debug_panel_trace_ctl.argtypes = [ctypes.c_int32, ctypes.c_void_p]
ctl_stream = torch.cuda.Stream()
buf = np.zeros(SLOTS * 32, dtype=np.uint64)


def record(on):
    assert lib.aps_debug_panel_trace_ctl(int(on), ctypes.c_void_p(ctl_stream.cuda_stream)) == 0


def read():
    used = lib.aps_debug_panel_trace(buf.ctypes.data, buf.nbytes)
    assert used >= 0
    return buf.reshape(SLOTS, 32)[:min(used, SLOTS)].astype(np.int64), used


_, d = bench.build_joint(dev, 0, P, 1)
net, wavs, lens = d["net"], d["wavs"], d["lens"]
net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
record(0)
with torch.no_grad():
    for b in range(2):
        net(wavs[b], lens)
    torch.cuda.synchronize()
    # (every launch issued from here on -- the eager reference passes and the captures -- gets slots: reset right
    # before the captures would need a hook; the eager passes' slots are simply never stamped again)
    assert lib.aps_debug_panel_trace_filter(0, 0, 1) == 0
    reps = PipelinedReplicas([lambda b=b: net(wavs[b], lens) for b in range(P)], workers=3, lstm_share=2)
print("stages of a batch:", reps.kinds[0], "| form", os.environ.get("APS_PANEL_FORM", "e (default)"))


def stage_b_only(i, stream):
    with torch.cuda.stream(stream):
        reps.pipelines[i][-1][0].replay()


def run(mode, rounds):
    """`rounds` passes over the resident batches in `mode`; recording is switched on behind the first pass and the
    trace keeps every launch node's LAST replay"""
    def one(i):
        if mode == "alone":
            stage_b_only(0, reps.streams[0])
        elif mode == "no_lstm":
            stage_b_only(i % P, reps.streams[i % 3])
        else:
            reps.submit(after_caller=False)
    for i in range(P):
        one(i)
    record(1)
    for i in range(P, rounds * P):
        one(i)
    reps.synchronize()
    record(0)


SHAPES = [(512, 512), (1024, 512), (1536, 512), (512, 1024)]
for mode in ("alone", "no_lstm", "pipeline"):
    assert lib.aps_debug_panel_trace_zero() == 0   # the stamps, not the slot assignment: the graphs keep theirs
    run(mode, 6)
    t_all, used = read()
    done = (t_all[:, 27] > t_all[:, 0]) & (t_all[:, 0] > 0)
    t_all = t_all[done]
    print(f"==== {mode}: {used} workgroup slots handed out, {len(t_all)} stamped")
    for n, k in SHAPES:
        nch = (k + 127) // 128
        t = t_all[(t_all[:, 28] >> 32 == n) & ((t_all[:, 28] & 0xffffffff) == k)]
        if not len(t):
            print(f"  N={n} K={k}: nothing recorded")
            continue
        life = t[:, 27] - t[:, 0]
        print(f"  N={n} K={k} ({nch} chunks of 128), {len(t)} workgroups, LN in {int(((t[:, 31] >> 8) & 1).sum())}: "
              f"life mean {life.mean():.0f} cycles (p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f})")
        split = mfma = bar = 0
        for c in range(nch):
            first = 1 if c == 0 else 4 + 3 * (c - 1)
            split = split + (t[:, 2 + 3 * c] - t[:, first])
            bar = bar + (t[:, 3 + 3 * c] - t[:, 2 + 3 * c])
            mfma = mfma + (t[:, 4 + 3 * c] - t[:, 3 + 3 * c])
        for name, v in (("entry -> first rows arrived", t[:, 1] - t[:, 0]), (f"{nch} x maxima + split + LDS writes", split),
                        (f"{nch} x barrier A", bar), (f"{nch} x MFMA loop + fold + barrier B", mfma),
                        ("last barrier -> epilogue (wide check, stats)", t[:, 26] - t[:, 4 + 3 * (nch - 1)]),
                        ("epilogue (+ prefetch, stores drained)", t[:, 27] - t[:, 26])):
            print(f"      {name:46s} mean {v.mean():8.0f}   p10 {np.percentile(v, 10):8.0f}   p90 {np.percentile(v, 90):8.0f}")
        # the launch as a whole: first entry -> last exit of the workgroups that share a C pointer and a launch
        # (a launch = one C pointer x one slot range; slots are contiguous per launch: split at gaps of C)
        key = t[:, 29] * 16 + ((t[:, 30] >> 32) & 0xf)   # (s_memtime is per XCD: spans are taken inside one)
        spans = {}
        for kk in np.unique(key):
            g = t[key == kk]
            spans.setdefault(int(kk) // 16, []).append(g[:, 27].max() - g[:, 0].min())
        worst = [max(v) for v in spans.values()]
        print(f"      launch span (first entry -> last exit on the slowest XCD): mean {np.mean(worst):.0f} cycles over "
              f"{len(worst)} launches; a wave's MFMAs occupy its SIMD for {(k // 32) * 6 * 32} cycles")
reps.close()
