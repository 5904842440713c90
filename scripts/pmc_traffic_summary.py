#!/usr/bin/env python
"""Per-kernel average FETCH_SIZE / WRITE_SIZE (KiB per dispatch, raw: see profiles/pmc_traffic.json
for the gfx950 corrections) from two rocprofv3 --pmc passes.
   python scripts/pmc_traffic_summary.py <fetch csv> <write csv>"""
import collections
import csv
import sys


def load(path):
    acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0]
        acc[name] += float(row["Counter_Value"])
        cnt[name] += 1
    return {k: (acc[k] / cnt[k], cnt[k]) for k in acc}


def main(fetch_csv, write_csv):
    f, w = load(fetch_csv), load(write_csv)
    print("kernel,dispatches,fetch_kb_raw_per_dispatch,write_kb_raw_per_dispatch")
    for name in sorted(f, key=lambda k: -f[k][0]):
        if not name.startswith(("void aps::", "aps::")):
            continue
        print(f'"{name}",{f[name][1]},{f[name][0]:.1f},{w.get(name, (0, 0))[0]:.1f}')


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
