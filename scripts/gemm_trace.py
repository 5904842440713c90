#!/usr/bin/env python
"""Where a K step of gemm_fp16x2_kernel spends its cycles: s_memtime stamps of one lane per wave of
eight workgroups (a library built with -DAPS_FP16X2_TRACE, selected through APS_AMD_LIB).

    APS_AMD_LIB=aps_amd/csrc/libaps_amd_trace.so python scripts/gemm_trace.py [M N K] [ln]

Stamps per K step: 0 top | 1 global requests issued | 2 MFMAs of the first half issued (waits for the
A fragments from LDS and the weight fragments) | 3 second half | 4 next A tile split and written to
LDS (waits for its global request) | 5 own LDS writes done | 6 through the barrier."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aps_amd import _native, nn_ops  # noqa: E402

args = [a for a in sys.argv[1:] if a != "ln"]
M, N, K = (int(a) for a in args[:3]) if len(args) >= 3 else (8064, 1024, 512)
use_ln = "ln" in sys.argv
dev = torch.device("cuda:0")
lib = _native.load()
nn_ops.SPLIT_MODE, nn_ops.SPLIT_LAYOUT = "1", 2
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).to(dev)
w = torch.nn.Parameter((torch.randn(N, K, generator=g) / K**0.5).to(dev), requires_grad=False)
ln = torch.nn.LayerNorm(K).to(dev) if use_ln else None
with torch.no_grad():
    for _ in range(5):
        y = nn_ops.linear(x, w, ln=ln)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = nn_ops.linear(x, w, ln=ln)
    e1.record()
    torch.cuda.synchronize()
print(f"M={M} N={N} K={K} ln={use_ln}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (incl. row_exp, traced build)")
buf = np.zeros(8 * 4 * 64 * 8, dtype=np.uint64)
lib.aps_debug_fp16x2_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
rc = lib.aps_debug_fp16x2_trace(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
t = buf.reshape(8, 4, 64, 8).astype(np.int64)
steps = (K + 31) // 32
names = ["issue loads", "MFMA half 1 (+waits)", "MFMA half 2 (+W re-request)", "A tile to LDS (+wait A)", "lgkmcnt(0)", "barrier"]
for slot in range(8):
    tt = t[slot, :, :steps, :7]
    if tt[0, 0, 0] == 0:
        continue
    d = np.diff(tt, axis=-1)          # [wave, step, 6]
    step_len = tt[:, 1:, 0] - tt[:, :-1, 0]
    print(f"workgroup slot {slot}: kernel entry->last barrier {int(tt[:, -1, 6].max() - tt[:, 0, 0].min())} cycles; "
          f"K step {step_len.mean():.0f} cycles (min {step_len.min()}, max {step_len.max()})")
    for k, nm in enumerate(names):
        print(f"    {nm:28s} mean {d[:, :, k].mean():7.0f}  waves {np.round(d[:, :, k].mean(1)).astype(int).tolist()}")
print("(s_memtime counts at the shader clock; 12 MFMAs of a K step occupy a SIMD for 384 cycles)")
# ---- the launch as a whole: every workgroup's entry / loop start / loop end / exit and where it ran
tiles = ((M + 63) // 64) * ((N + 127) // 128)
wg = np.zeros(4096 * 8, dtype=np.uint64)
lib.aps_debug_fp16x2_wgtrace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert lib.aps_debug_fp16x2_wgtrace(wg.ctypes.data, wg.nbytes) == 0
wg = wg.reshape(4096, 8)[:min(tiles, 4096)].astype(np.int64)
hw, xcc = wg[:, 2], wg[:, 3] & 0xf
cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)   # (xcc, se, sh, cu)
pro, loop, epi = wg[:, 5] - wg[:, 4], wg[:, 6] - wg[:, 5], wg[:, 7] - wg[:, 6]
print(f"{len(wg)} workgroups on {len(set(cu.tolist()))} CUs; per workgroup, cycles: prologue {pro.mean():.0f} "
      f"(p10 {np.percentile(pro, 10):.0f}, p90 {np.percentile(pro, 90):.0f}), K loop {loop.mean():.0f} "
      f"({np.percentile(loop, 10):.0f} .. {np.percentile(loop, 90):.0f}), epilogue {epi.mean():.0f} "
      f"({np.percentile(epi, 10):.0f} .. {np.percentile(epi, 90):.0f})")
for x in sorted(set(xcc.tolist())):
    sel = xcc == x
    t0 = wg[sel, 4].min()
    span = wg[sel, 7].max() - t0
    starts = np.sort(wg[sel, 4] - t0)
    per_cu = {}
    for c, a, b in zip(cu[sel], wg[sel, 4] - t0, wg[sel, 7] - t0):
        per_cu.setdefault(int(c), []).append((int(a), int(b)))
    busy = np.mean([sum(b - a for a, b in v) / span for v in per_cu.values()])
    n_per = [len(v) for v in per_cu.values()]
    late = int((starts > 2000).sum())
    print(f"  XCC {x}: {int(sel.sum())} workgroups on {len(per_cu)} CUs ({min(n_per)}..{max(n_per)} per CU), "
          f"span {span} cycles, mean resident workgroups per CU {busy:.2f}, {late} started later than 2000 "
          f"cycles after the first (starts p50 {np.percentile(starts, 50):.0f}, p90 {np.percentile(starts, 90):.0f}, "
          f"max {starts.max()})")
one = sorted(per_cu.items())[0]
print("  one CU's workgroups (entry, exit):", sorted(one[1]))
