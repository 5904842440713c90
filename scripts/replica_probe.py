"""Where the replay runs matters: the joint step captured once, replayed on torch's default (null)
stream and on a created stream; then R GraphReplicas.  python scripts/replica_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from aps_amd.replicas import GraphReplicas  # noqa: E402


def timed(launch, steps=200):
    for _ in range(10):
        launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        launch()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def main():
    device = torch.device("cuda:0")
    _, dev = bench.build_joint(device, 0)
    net, wav, lens = dev["net"], dev["wav"], dev["lens"]
    net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
    with torch.no_grad():
        for _ in range(3):
            net(wav, lens)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            net(wav, lens)
        side = torch.cuda.Stream()

        def on_side():
            with torch.cuda.stream(side):
                g.replay()
        print(f"one graph, default stream : {timed(g.replay):.3f} ms", flush=True)
        print(f"one graph, created stream : {timed(on_side):.3f} ms", flush=True)
        print(f"one graph, default stream : {timed(g.replay):.3f} ms", flush=True)
        for R in (1, 2):
            reps = GraphReplicas(lambda: net(wav, lens), replicas=R)
            print(f"GraphReplicas R={R}          : {timed(reps.submit):.3f} ms", flush=True)
            if R == 1:
                print(f"  its graph on the default stream: {timed(reps.graphs[0].replay):.3f} ms",
                      flush=True)
            del reps


if __name__ == "__main__":
    main()
