#!/usr/bin/env python
"""Where a workgroup of gemm_panel_kernel spends its life: s_memtime stamps of every wave of the first
2048 workgroups (a library built with -DAPS_PANEL_TRACE, selected through APS_AMD_LIB).

    scripts/build_variant_lib.sh ptrace gemm_panel gemm_panel.hip -DAPS_PANEL_TRACE
    APS_AMD_LIB=aps_amd/csrc/libaps_amd_ptrace.so python scripts/panel_trace.py M N K form [ln]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aps_amd import _native, nn_ops  # noqa: E402

M, N, K, form = (int(a) for a in sys.argv[1:5])
use_ln = "ln" in sys.argv
dev = torch.device("cuda:0")
lib = _native.load()
nn_ops.SPLIT_MODE, nn_ops.SPLIT_LAYOUT, nn_ops.PANEL_FORM = "1", 3, form
g = torch.Generator().manual_seed(0)
xs = [torch.randn(M, K, generator=g).to(dev) for _ in range(4)]
w = torch.nn.Parameter((torch.randn(N, K, generator=g) / K**0.5).to(dev), requires_grad=False)
r = torch.randn(M, N, generator=g).to(dev)
ln = torch.nn.LayerNorm(K).to(dev) if use_ln else None
with torch.no_grad():
    for i in range(6):
        y = nn_ops.linear(xs[i % 4], w, residual=r, ln=ln)
    torch.cuda.synchronize()
buf = np.zeros(2048 * 8 * 16, dtype=np.uint64)
lib.aps_debug_panel_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert lib.aps_debug_panel_trace(buf.ctypes.data, buf.nbytes) == 0
rows, cols = lib.aps_linear_panel_rows(M, N, form), lib.aps_linear_panel_cols(M, N, form)
nw = cols // 32
tiles = min(2048, ((M + rows - 1) // rows) * ((N + cols - 1) // cols))
t = buf.reshape(2048, 8, 16)[:tiles, :nw].astype(np.int64)
chunk = 256 if form == 1 else 128
nch = min(4, (K + chunk - 1) // chunk)
start = t[:, :, 0].min()
print(f"M={M} N={N} K={K} form {form} ({rows} x {cols}, {nw} waves, chunks of {chunk}) ln={use_ln}: {tiles} workgroups traced")
print(f"  launch: first entry -> last exit {int(t[:, :, 15].max() - start)} cycles; entries spread over "
      f"{int(t[:, :, 0].max() - start)}; workgroup life mean {np.mean(t[:, :, 15].max(1) - t[:, :, 0].min(1)):.0f}")
ph = [("entry -> first rows arrived", 0, 1)]
for c in range(nch):
    first = 1 if c == 0 else 4 + 3 * (c - 1)
    ph += [(f"chunk {c}: maxima + split + LDS writes", first, 2 + 3 * c), (f"chunk {c}: barrier A", 2 + 3 * c, 3 + 3 * c),
           (f"chunk {c}: MFMA loop + fold + barrier B", 3 + 3 * c, 4 + 3 * c)]
ph += [("last barrier -> epilogue begins (wide check, stats)", 4 + 3 * (nch - 1), 14), ("epilogue (+ prefetch, stores drained)", 14, 15)]
for name, a, b in ph:
    d = t[:, :, b] - t[:, :, a]
    print(f"  {name:52s} mean {d.mean():8.0f}   p10 {np.percentile(d, 10):8.0f}   p90 {np.percentile(d, 90):8.0f}")
xcc = (buf.reshape(2048, 8, 16)[:tiles, 0, 13] >> np.uint64(32)).astype(np.int64) & 0xf
print("  workgroups per XCC:", np.bincount(xcc, minlength=8).tolist())
mf = (K // 32) * 6 * (rows // 32) * 32
print(f"  (a wave's MFMAs occupy its SIMD for {mf} cycles in all; s_memtime ticks at the shader clock)")
