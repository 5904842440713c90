#!/usr/bin/env python
"""Counters of one kernel (substring match) averaged per launch, from rocprofv3 --pmc csv files:
   python scripts/pmc_kernel_table.py <substring> <dir>/p_counter_collection.csv [...]"""
import collections
import csv
import sys

sub = sys.argv[1]
acc = collections.defaultdict(float)
n = collections.defaultdict(set)
for path in sys.argv[2:]:
    for row in csv.DictReader(open(path)):
        if sub in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
            n[row["Counter_Name"]].add(row["Dispatch_Id"])
for k in sorted(acc):
    print(f"{k:32s} {acc[k] / max(len(n[k]), 1):16.0f}  ({len(n[k])} launches)")
