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
    # graph-captured chains of 20 dependent GEMMs: device time per GEMM without host launch overhead
    for M, N, K in ((2016, 512, 512), (2016, 1024, 512), (2016, 512, 1024), (2016, 1536, 512), (64, 64, 512)):
        xs = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
        w2 = torch.randn(K, N, device="cuda")
        linear(linear(xs, w, b), w2); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = xs
            for _ in range(10):
                y = linear(linear(y, w, b), w2)   # N-wide then back to K-wide: two shapes per pair
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"graph chain {M}x{N}x{K} + {M}x{K}x{N}: {e0.elapsed_time(e1) / 100 * 1e3:6.1f} us per pair")
