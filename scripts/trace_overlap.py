#!/usr/bin/env python
"""Summary of a rocprofv3 kernel trace (csv) of several streams: per kernel calls / mean duration / share of the
wall time it covers, per queue busy share, and how many kernels run at once on average.
    python scripts/trace_overlap.py <..._kernel_trace.csv> [skip_fraction]
(the first `skip_fraction` of the trace, warm-up and capture, is left out; default 0.5)"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
              r.get("Stream_Id", "?")) for r in rows), key=lambda e: e[0])
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t0 + skip * (t1 - t0)
ev = [e for e in ev if e[0] >= cut]
t0, t1 = ev[0][0], max(e[1] for e in ev)
wall = t1 - t0
print(f"window {wall / 1e6:.2f} ms, {len(ev)} kernels")
per = defaultdict(lambda: [0, 0])
queue = defaultdict(int)
for s, e, n, q, st in ev:
    k = n.split("(")[0][-70:]
    per[k][0] += 1
    per[k][1] += e - s
    queue[(q, st)] += e - s
print("kernel, calls, mean us, summed share of the window")
for k, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {k:70s} {c:6d} {d / c / 1e3:9.2f} {d / wall:6.3f}")
print("queue/stream busy share:")
for k, d in sorted(queue.items(), key=lambda kv: -kv[1]):
    print(f"  {k} {d / wall:6.3f}")
# concurrency: sweep
pts = []
for s, e, *_ in ev:
    pts.append((s, 1))
    pts.append((e, -1))
pts.sort()
level, last, hist = 0, pts[0][0], defaultdict(int)
for t, d in pts:
    hist[level] += t - last
    level += d
    last = t
print("kernels running at once -> share of the window:", {k: round(v / wall, 3) for k, v in sorted(hist.items())})
