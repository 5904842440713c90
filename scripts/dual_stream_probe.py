#!/usr/bin/env python
"""Does running the encoder on two half batches concurrently (two streams inside one graph) beat
one full-batch pass?  (latency-bound M = 2016 GEMMs: the halves' fixed phases may overlap)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def timed_graph(fn, reps=30):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    cpu, dev = bench.build_joint(torch.device("cuda", 0), 0)
    net = dev["net"]
    enc = net.asr
    x = torch.randn(32, 249, 80, device="cuda")
    n = torch.full((32,), 249, device="cuda", dtype=torch.int64)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def full():
        return enc(x, n)

    def halves():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            a = enc(x[:16], n[:16])
        with torch.cuda.stream(s2):
            b = enc(x[16:], n[16:])
        cur.wait_stream(s1)
        cur.wait_stream(s2)
        return a, b

    def halves_serial():
        return enc(x[:16], n[:16]), enc(x[16:], n[16:])

    print(f"full batch 32        : {timed_graph(full):.3f} ms")
    print(f"2 x 16, one stream   : {timed_graph(halves_serial):.3f} ms")
    print(f"2 x 16, two streams  : {timed_graph(halves):.3f} ms")
