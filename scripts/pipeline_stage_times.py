#!/usr/bin/env python
"""Stage durations of the pipelined joint step UNDER LOAD (rocprofv3 serialises the queues, so its trace cannot show
them): the bench's model and batches on PipelinedReplicas, submissions replayed with timing events around every
stage -- mean duration of A (front), L (LSTM stack), B (the rest), how busy the head stream and the workers are.
    python scripts/pipeline_stage_times.py [workers] [share] [front]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from aps_amd.replicas import PipelinedReplicas  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 3
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
FRONT = sys.argv[3] if len(sys.argv) > 3 else "head"
P, ROUNDS = 12, 8
dev = torch.device("cuda:0")
_, d = bench.build_joint(dev, 0, P, 1)
net, wavs, lens = d["net"], d["wavs"], d["lens"]
net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
with torch.no_grad():
    for b in range(2):
        net(wavs[b], lens)
    torch.cuda.synchronize()
    reps = PipelinedReplicas([lambda b=b: net(wavs[b], lens) for b in range(P)], workers=W, lstm_share=S, front=FRONT)


def ev():
    return torch.cuda.Event(enable_timing=True)


def submit(i, log):
    worker = reps.streams[i % reps.workers]
    prev, marks = None, []
    for k, (graph, _) in enumerate(reps.pipelines[i]):
        st = reps.lstm_stream if reps.kinds[i][k] in ("l", "m") else worker
        if k == 0 and reps.front_stream is not None:
            st = reps.front_stream
            if reps._done[i] is not None:
                st.wait_event(reps._done[i])
        if prev is not None:
            st.wait_event(prev)
        with torch.cuda.stream(st):
            a = ev()
            a.record(st)
            graph.replay()
            prev = ev()
            prev.record(st)
        marks.append((a, prev))
    reps._done[i] = prev
    log.append(marks)


for i in range(2 * P):
    submit(i % P, [])
reps.synchronize()
log = []
t0 = time.perf_counter()
for i in range(ROUNDS * P):
    submit(i % P, log)
reps.synchronize()
step_ms = 1e3 * (time.perf_counter() - t0) / (ROUNDS * P)
kinds = reps.kinds[0]
names = [{"a": "A (front)", "l": "L (LSTM stack)", "m": "M (front end's tail)", "b": "B (encoder)"}[k] for k in kinds]
dur = [[a.elapsed_time(b) for a, b in (m[k] for m in log[P:])] for k in range(len(kinds))]
print(f"workers {W}, lstm_share {S}, front {FRONT}: {step_ms:.3f} ms per step ({32 / step_ms * 1e3:.0f} utt/s)")
for k in range(len(kinds)):
    x = sorted(dur[k])
    print(f"  {names[k]:16s} mean {sum(x) / len(x):.3f} ms   p10 {x[len(x) // 10]:.3f}   p90 {x[9 * len(x) // 10]:.3f}")
mean = [sum(x) / len(x) for x in dur]
on_head = [k in ("l", "m") or (k == "a" and FRONT == "head") for k in kinds]
head = sum(m for m, h in zip(mean, on_head) if h)
work = sum(m for m, h in zip(mean, on_head) if not h)
print(f"  head stream busy {head / step_ms:.2f} of the time; a worker {work / (W * step_ms):.2f}")
gap = [m[-2][1].elapsed_time(m[-1][0]) for m in log[P:]]
print(f"  last head stage done -> the worker's stage starts: mean {sum(gap) / len(gap):.3f} ms")
reps.close()
