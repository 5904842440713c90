#!/usr/bin/env python
"""The persistent LSTM-stack launch of the joint step (stage L) with k encoder stages (stage B: one
aps_conformer_stack launch each) looping beside it on k worker streams, k = 0 .. 6, and with the front-end stages
beside it: ms per LSTM launch (HIP events on the head stream) -- what the head stream of the pipeline pays for its
neighbours.   python scripts/lstm_under_load_probe.py"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from aps_amd.replicas import PipelinedReplicas  # noqa: E402

W, P = 6, 12
dev = torch.device("cuda:0")
_, d = bench.build_joint(dev, 0, P, 1)
net, wavs, lens = d["net"], d["wavs"], d["lens"]
net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
with torch.no_grad():
    for b in range(2):
        net(wavs[b], lens)
    torch.cuda.synchronize()
    reps = PipelinedReplicas([lambda b=b: net(wavs[b], lens) for b in range(P)], workers=W, lstm_share=2,
                             front="worker", mid="worker", lookahead=True)
kinds = reps.kinds[0]


def stage(i, kind, stream):
    with torch.cuda.stream(stream):
        reps.pipelines[i][kinds.index(kind)][0].replay()


def run(name, neighbours, rounds=30):
    """neighbours: list of (kind, stream) looping beside the LSTM launches"""
    torch.cuda.synchronize()
    evs = []
    n_side = 0
    t0 = time.perf_counter()
    for r in range(rounds):
        for j, (kind, st) in enumerate(neighbours):
            stage((r * len(neighbours) + j) % P, kind, st)
            n_side += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(reps.lstm_stream):
            e0.record()
        stage(r % P, "l", reps.lstm_stream)
        with torch.cuda.stream(reps.lstm_stream):
            e1.record()
        evs.append((e0, e1))
        if r % 4 == 3:   # keep the side streams a few launches deep, not unbounded
            reps.lstm_stream.synchronize()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = sorted(a.elapsed_time(b) for a, b in evs[5:])
    print(f"{name:58s} LSTM launch median {ms[len(ms) // 2]:.3f} ms (p10 {ms[len(ms) // 10]:.3f}, p90 {ms[9 * len(ms) // 10]:.3f})"
          f"   wall {1e3 * wall / rounds:.3f} ms per round", flush=True)


st = reps.streams
run("alone", [])
for k in (1, 2, 3, 4, 5, 6):
    run(f"{k} encoder stage(s) per LSTM launch on {k} worker stream(s)", [("b", st[j]) for j in range(k)])
run("6 front ends (stage A) per LSTM launch on 6 streams", [("a", st[j]) for j in range(6)])
run("1 front end (stage A) + 1 tail (stage M) per LSTM launch", [("a", st[0]), ("m", st[1])])
run("as the pipeline: 1 A + 1 M + 1 B per LSTM launch on 3 of 6 streams", [("a", st[0]), ("m", st[1]), ("b", st[2])])
reps.close()
