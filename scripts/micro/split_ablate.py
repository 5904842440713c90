#!/usr/bin/env python
"""Steady-state cost of the parts of gemm_split_swp_kernel: csrc/gemm_split.hip is compiled with
-DAPS_SPLIT_ABLATE=<mask> (1 global requests, 2 LDS writes, 4 split VALU, 8 barrier, 16 operand
fetches, 32 MFMAs left out of the loop; results are garbage, only the time matters) and timed at
M = 8064, N = 512, K = 2048 / 512.   python scripts/micro/split_ablate.py [build|run]"""
import ctypes
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
MASKS = [0, 1, 2, 3, 8, 16, 32, 3 | 8, 3 | 16, 3 | 8 | 16, 32 | 16, 32 | 1]


def so(mask):
    return os.path.join(here, f"split_abl_{mask}.so")


def build():
    procs = []
    for m in MASKS:
        procs.append(subprocess.Popen(
            ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
             "-mcode-object-version=5", "-Wno-unused-value", f"-DAPS_SPLIT_ABLATE={m}",
             "-I", os.path.join(root, "aps_amd", "csrc"), "-o", so(m),
             os.path.join(root, "aps_amd", "csrc", "gemm_split.hip")]))
    for p in procs:
        assert p.wait() == 0


def run():
    import torch
    sys.path.insert(0, root)
    from scripts.r02_probe import graph_time
    os.environ["APS_SPLIT_KERNEL"] = os.environ.get("APS_SPLIT_KERNEL", "pc")
    torch.manual_seed(0)
    M, N = 8064, 512
    for m in MASKS:
        lib = ctypes.CDLL(so(m))
        lib.aps_linear_split_size.restype = ctypes.c_int64
        lib.aps_linear_split_size.argtypes = [ctypes.c_int64] * 2
        P, I64 = ctypes.c_void_p, ctypes.c_int64
        lib.aps_linear_split_weight.argtypes = [P, P, I64, I64, I64, ctypes.c_int32, P]
        lib.aps_linear_split.argtypes = [P] * 6 + [I64] * 5 + [ctypes.c_int32, ctypes.c_float,
                                                              ctypes.c_float, ctypes.c_int32, P]
        row = []
        for K in (512, 2048):
            x = torch.randn(M, K, device="cuda")
            w = torch.randn(N, K, device="cuda")
            planes = torch.empty(lib.aps_linear_split_size(N, K) // 2, device="cuda", dtype=torch.int16)
            lib.aps_linear_split_weight(w.data_ptr(), planes.data_ptr(), N, K, K, 0, None)
            out = torch.empty(M, N, device="cuda")
            st = torch.cuda.current_stream().cuda_stream

            def fn():
                lib.aps_linear_split(x.data_ptr(), planes.data_ptr(), None, None, None, out.data_ptr(),
                                     M, N, K, K, N, 0, 1.0, 0.0, 0, torch.cuda.current_stream().cuda_stream)
            row.append(graph_time(fn))
        step = (row[1] - row[0]) / 48
        print(f"ablate {m:2d}: K=512 {row[0]:6.1f} us  K=2048 {row[1]:6.1f} us  per step {step * 1e3:5.0f} ns "
              f"= {step * 2.4e3:5.0f} cycles @2.4GHz", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "both"
    if what in ("build", "both"):
        build()
    if what in ("run", "both"):
        run()
