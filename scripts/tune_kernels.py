#!/usr/bin/env python
"""Kernel-level timing on the GPU box: each stage of the config-2 step, HIP events, many reps."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def timeit(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3  # median, min (us)


def main():
    dev = torch.device("cuda:0")
    cpu, w = bench.build_workload(dev, 0)
    st = bench.Stages(w)
    enh = w["enh"]
    enh.nan_policy = "off"
    with torch.no_grad():
        for var in sys.argv[1:] or ["2"]:
            os.environ["APS_STFT_ITERS"] = var
            med, mn = timeit(lambda: enh.forward_stft.to_store(w["x"]))
            print(f"stft only iters={var}: median {med:7.1f} us  min {mn:7.1f} us")
        os.environ.pop("APS_STFT_ITERS", None)
        st.step()


        for name in bench.Stages.ORDER:
            med, mn = timeit(lambda: st.run_stage(name))
            gbs = bench.ALGO_BYTES[name] * bench.BATCH / (med * 1e-6) / 1e9
            print(f"{name:17s}: median {med:7.1f} us  min {mn:7.1f} us  {gbs:7.0f} GB/s algo")


if __name__ == "__main__":
    main()
