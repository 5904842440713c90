#!/usr/bin/env python
"""How much do the kernels of two batches in flight really overlap?  From a rocprofv3 kernel trace
(trace_kernel_trace.csv): time with >= 1 and >= 2 kernels running, per kernel family the share of its
run time spent beside a kernel of the OTHER queue, and beside which family.
    python scripts/trace_overlap.py <kernel_trace.csv> [skip_fraction]"""
import collections
import csv
import sys


def family(name):
    for key in ("gemm_panel", "gemm_fp16x2", "fp16x2_split", "gemm_f32", "lstm", "attention_small", "glu_dwconv",
                "conv_", "stft", "features", "covariance", "beamform", "spin_kernel"):
        if key in name:
            return key
    return "other"


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]          # the timed regions are at the end
rows = [r for r in rows if "spin_kernel" not in r["Kernel_Name"]]
ev = []
for i, r in enumerate(rows):
    ev.append((int(r["Start_Timestamp"]), 1, i))
    ev.append((int(r["End_Timestamp"]), 0, i))
ev.sort()
active, last = set(), ev[0][0]
busy1 = busy2 = 0
fam_time = collections.Counter()
fam_beside = collections.Counter()
pair = collections.Counter()
for t, kind, i in ev:
    dt = t - last
    if dt > 0 and active:
        busy1 += dt
        qs = {rows[j]["Queue_Id"] for j in active}
        if len(active) >= 2:
            busy2 += dt
        for j in active:
            f = family(rows[j]["Kernel_Name"])
            fam_time[f] += dt
            others = [k for k in active if rows[k]["Queue_Id"] != rows[j]["Queue_Id"]]
            if others:
                fam_beside[f] += dt
                for k in others:
                    pair[(f, family(rows[k]["Kernel_Name"]))] += dt
    last = t
    if kind:
        active.add(i)
    else:
        active.discard(i)
span = ev[-1][0] - ev[0][0]
print(f"{len(rows)} kernels over {span / 1e6:.2f} ms: >= 1 running {busy1 / 1e6:.2f} ms ({busy1 / span:.2f}), "
      f">= 2 running {busy2 / 1e6:.2f} ms ({busy2 / span:.2f}); sum of durations {sum(fam_time.values()) / 1e6:.2f} ms")
print(f"{'family':18s} {'run ms':>8s} {'beside other queue':>20s}   mostly beside")
for f, t in fam_time.most_common():
    best = sorted(((v, k[1]) for k, v in pair.items() if k[0] == f), reverse=True)[:3]
    print(f"{f:18s} {t / 1e6:8.2f} {fam_beside[f] / max(t, 1):20.2f}   " +
          ", ".join(f"{k} {v / max(t, 1):.2f}" for v, k in best))
