"""HBM rates of the streaming kernels outside the bench workloads: FixedBeamformer, DfTransform,
speed perturbation, SpecAugment at the benchmark geometry (32 utterances, 4 s, 257 bins, 249
frames).  Algorithmic bytes = every input element read once + every output element written once."""
import os
import random
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aps_amd.transform import DfTransform, FixedBeamformer  # noqa: E402
from aps_amd.transform.asr import SpecAugTransform, SpeedPerturbTransform  # noqa: E402


def timed(fn, reps=50):
    """device time per call: the stream is held busy by a spin kernel while the host enqueues the
    calls (the Python wrappers cost more host time than these kernels run), as in bench.py"""
    import bench
    for _ in range(5):
        fn()
    th.cuda.synchronize()
    bench.hold_stream(bench.spin_cycles_for(40.0))
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    dev = th.device("cuda:0")
    N, T, F = 32, 249, 257
    with th.no_grad():
        for B, C in ((8, 4), (16, 8)):
            bf = FixedBeamformer(B, C, F).to(dev)
            r, i = th.randn(N, C, F, T, device=dev), th.randn(N, C, F, T, device=dev)
            s = timed(lambda: bf(r, i))
            byt = 4 * 2 * N * F * T * (C + B)
            print(f"fixed beamformer B={B} C={C}: {s * 1e6:.1f} us, {byt / s / 1e12:.2f} TB/s algorithmic")
        p = th.rand(N, 7, T, F, device=dev)
        doa = th.rand(N, device=dev)
        for D in (1, 8):
            df = DfTransform(num_bins=F, num_doas=D).to(dev)
            s = timed(lambda: df(p, doa))
            byt = 4 * N * T * F * (7 + D)
            print(f"directional feature D={D}: {s * 1e6:.1f} us, {byt / s / 1e12:.2f} TB/s algorithmic")
        wav = th.randn(N, 64000, device=dev)
        sp = SpeedPerturbTransform().to(dev).train()
        s = timed(lambda: sp(wav))
        print(f"speed perturbation 32 x 64000: {s * 1e6:.1f} us (host draw + launch), "
              f"{4 * N * 64000 * 2 / s / 1e12:.2f} TB/s algorithmic")
        x = th.randn(N, T, 80, device=dev)
        random.seed(0)
        for zero in (True, False):
            aug = SpecAugTransform(p=1.0, mask_zero=zero).to(dev).train()
            s = timed(lambda: aug(x))
            print(f"spec augment mask_zero={zero} 32 x 249 x 80: {s * 1e6:.1f} us (host draws + "
                  f"band upload + launch)")


if __name__ == "__main__":
    main()
