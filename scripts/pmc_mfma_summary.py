#!/usr/bin/env python
"""Per-kernel MFMA busy fraction from a rocprofv3 --pmc run
   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace \\
       --output-format csv -d <dir> -o p -- python bench.py --eager --steps 5 --warmup 2
   python scripts/pmc_mfma_summary.py <dir>/p_counter_collection.csv > profiles/rNN_joint_pmc_mfma.csv
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCCs x 1024 SIMDs): the share of
SIMD-cycles of the launches' active time in which the matrix pipe was busy."""
import collections
import csv
import sys


def main(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0]
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
        launches[name].add(row["Dispatch_Id"])
    print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES "
          "--kernel-trace -- python bench.py --eager --steps 5 --warmup 2")
    print("# counters summed over the launches of a kernel; mfma_busy_frac = MFMA_BUSY / "
          "(GRBM_GUI_ACTIVE / 8 XCC x 1024 SIMDs)")
    print("kernel,launches,GRBM_GUI_ACTIVE,SQ_BUSY_CU_CYCLES,SQ_VALU_MFMA_BUSY_CYCLES,SQ_WAVES,"
          "mfma_busy_frac")
    rows = sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))
    for name, c in rows[:12]:
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        frac = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024) if gui else 0.0
        print(f'"{name}",{len(launches[name])},{gui:.0f},{c.get("SQ_BUSY_CU_CYCLES", 0):.0f},'
              f'{c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0):.0f},{c.get("SQ_WAVES", 0):.0f},{frac:.4f}')


if __name__ == "__main__":
    main(sys.argv[1])
