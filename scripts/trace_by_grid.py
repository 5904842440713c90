#!/usr/bin/env python
"""rocprofv3 kernel trace -> per (kernel, grid size) launch count and mean / total duration
   python scripts/trace_by_grid.py <dir>/trace_kernel_trace.csv [substring]"""
import collections
import csv
import sys

sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: [0, 0.0])
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Kernel_Name"].split("(")[0]
    if sub not in name:
        continue
    key = (name[-60:], row.get("Grid_Size", row.get("Grid_Size_X", "?")), row.get("Workgroup_Size", row.get("Workgroup_Size_X", "?")))
    a = acc[key]
    a[0] += 1
    a[1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in acc.values())
for key, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{key[0]:60s} grid {key[1]:>9s} wg {key[2]:>4s}  n {a[0]:6d}  mean {a[1] / a[0]:8.1f} us  total {a[1] / 1e3:8.2f} ms "
          f"({100 * a[1] / tot:4.1f} %)")
