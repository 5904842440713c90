import os, sys
sys.path.insert(0, "/root/repo")
import torch
from aps_amd.nn_ops import linear
def bench(M, N, K, reps=50):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    for _ in range(3): linear(x, w, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): linear(x, w, b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
with torch.no_grad():
    for M, N in ((2016, 512), (2016, 1024), (64, 64), (512, 512)):
        print(M, N, " ".join(f"K={K}: {bench(M, N, K):6.1f}us" for K in (32, 64, 128, 256, 512, 1024, 2048)))
    # graph-captured chain of 20 GEMMs: per-kernel time without host launch overhead
    x = torch.randn(2016, 512, device="cuda"); w = torch.randn(512, 512, device="cuda"); b = torch.randn(512, device="cuda")
    linear(x, w, b); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = x
        for _ in range(20): y = linear(y, w, b)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print("graph chain 2016x512x512 per GEMM us:", e0.elapsed_time(e1) / 200 * 1e3)
