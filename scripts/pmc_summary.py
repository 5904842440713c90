#!/usr/bin/env python
"""Per-kernel sums of whatever counters a rocprofv3 --pmc run collected (one CSV per pass; several passes are joined
by kernel name).   python scripts/pmc_summary.py <pass1>/p_counter_collection.csv [<pass2>/...] > profiles/rNN_x.csv
Counters are summed over the launches of a kernel; `launches` = dispatches seen in the first pass."""
import collections
import csv
import sys


def main(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    names = []
    for pi, path in enumerate(paths):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0]
            c = row["Counter_Name"]
            if c not in names:
                names.append(c)
            acc[k][c] += float(row["Counter_Value"])
            if pi == 0:
                launches[k].add(row["Dispatch_Id"])
    order = sorted(acc, key=lambda k: -max(acc[k].get("GRBM_GUI_ACTIVE", 0), acc[k].get("SQ_WAVE_CYCLES", 0),
                                           acc[k].get("SQ_BUSY_CYCLES", 0), len(launches[k])))
    print("kernel,launches," + ",".join(names))
    for k in order[:24]:
        print(f'"{k}",{len(launches[k])},' + ",".join(f"{acc[k].get(c, 0):.0f}" for c in names))


if __name__ == "__main__":
    main(sys.argv[1:])
