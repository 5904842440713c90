#!/usr/bin/env python
"""What does a RESIDENT kernel on every CU cost another stream's projections?  The two-plane GEMM launches of a
32-utterance joint step (recorded, as in gemm_sequence_overlap.py) are re-issued on one stream while a mostly idle
kernel of a chosen footprint (scripts/micro/occupier.hip: 256 workgroups x 256 threads, VGPRs 43 / 128 / 240,
LDS 0 / 70 KB, optional gather-like traffic) sits on another.
    python scripts/occupier_probe.py            (on an MI355X; build occupier.so first, see occupier.hip)"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from aps_amd import nn_ops  # noqa: E402


def main():
    occ = C.CDLL(os.path.join(ROOT, "scripts", "micro", "occupier.so"))
    occ.occupy.restype = C.c_int
    occ.occupy.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int,
                           C.c_int, C.c_void_p]
    dev = torch.device("cuda", 0)
    _, d = bench.build_joint(dev, 0, 2, 1)
    net, wavs, lens = d["net"], d["wavs"], d["lens"]
    net.enh_transform.nan_policy = net.asr_transform.nan_policy = "deferred"
    recs = []
    with torch.no_grad():
        for w in wavs:
            net(w, lens)
        torch.cuda.synchronize()
        for w in wavs:
            nn_ops.GEMM_RECORD = rec = []
            net(w, lens)
            torch.cuda.synchronize()
            nn_ops.GEMM_RECORD = None
            recs.append(rec)
    calls = [[c for c, _, _, _ in r] for r in recs]
    n = len(calls[0])
    gemm_stream, occ_stream = torch.cuda.Stream(), torch.cuda.Stream()
    sink = torch.zeros(4, device=dev)
    src = torch.randn(64 * 1024 * 1024, device=dev)  # 256 MB

    def issue(c, stream):
        fn, fargs = c.__defaults__
        fn(*fargs[:-1], stream.cuda_stream)

    def run(regs, blocks, lds, loads, sleep, reps=8):
        torch.cuda.synchronize()
        # the occupier covers the whole timed region (it ends by its own clock)
        est_ms = 2.6 * 2 * reps
        if blocks:
            occ.occupy(regs, blocks, lds, est_ms, sink.data_ptr(), src.data_ptr(), src.numel() * 4, loads, sleep,
                       occ_stream.cuda_stream)
            time.sleep(0.002)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(gemm_stream):
            e0.record()
            for _ in range(reps):
                for b in range(2):
                    for c in calls[b]:
                        issue(c, gemm_stream)
            e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (2 * reps)

    print(f"{n} two-plane GEMM launches per step; ms per batch of GEMMs on one stream")
    cases = [("alone", 0, 0, 0, 0, 0),
             ("256 WG,  43 VGPR,  0 KB LDS, idle", 24, 256, 0, 0, 8),
             ("256 WG, 128 VGPR,  0 KB LDS, idle", 100, 256, 0, 0, 8),
             ("256 WG, 240 VGPR,  0 KB LDS, idle", 210, 256, 0, 0, 8),
             ("256 WG,  43 VGPR, 70 KB LDS, idle", 24, 256, 70 * 1024, 0, 8),
             ("256 WG, 240 VGPR, 70 KB LDS, idle", 210, 256, 70 * 1024, 0, 8),
             ("256 WG,  43 VGPR,  0 KB LDS, 24 x 16 B per lane and ~3 us (the LSTM's gather volume)", 24, 256, 0, 24, 1),
             ("256 WG, 240 VGPR, 70 KB LDS, same traffic", 210, 256, 70 * 1024, 24, 1),
             ("128 WG, 240 VGPR, 70 KB LDS, idle", 210, 128, 70 * 1024, 0, 8),
             ("alone", 0, 0, 0, 0, 0)]
    for label, regs, blocks, lds, loads, sleep in cases:
        ms = [run(regs, blocks, lds, loads, sleep) for _ in range(2)]
        print(f"{label:90s} {ms[0]:.3f} {ms[1]:.3f}")


if __name__ == "__main__":
    main()
