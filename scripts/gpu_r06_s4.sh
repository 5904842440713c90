#!/bin/bash
# Round 6, visit 5: rocprof kernel durations (one stream, panel tiles only) for panel forms e and f -- the unperturbed
# per-launch truth the instrumented workgroup trace is compared with
set -u
R=$(pwd)
O=gpurun_out/r06_s4
mkdir -p $O
export TMPDIR=/tmp
for f in e f; do
  (cd /tmp && env APS_PANEL_FORM=$f APS_GEMM_KGROUP=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr_$f -o t -- \
     python $R/bench.py --group 1 --merged-group 0 --replicas 1 --pipeline 0 --steps 40 --warmup 5 --no-cpu-baseline > $R/$O/tr_$f.log 2>&1)
  fcsv=$(find $O/tr_$f -name "*kernel_stats.csv" | head -1)
  [ -n "$fcsv" ] && cp "$fcsv" $O/joint32_one_stream_form_${f}_kernel_stats.csv
  ktr=$(find $O/tr_$f -name "*kernel_trace.csv" | head -1)
  [ -n "$ktr" ] && python - "$ktr" <<'PY' > $O/panel_by_grid_$f.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "gemm_panel_kernel" in n or "gemm_kgroup" in n:
        acc[(n.split("<")[1].split(">")[0], r.get("Grid_Size") or r.get("Grid_Size_X"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v.sort()
    print(k, "launches", len(v), "median us %.2f mean %.2f p10 %.2f p90 %.2f" % (v[len(v)//2], sum(v)/len(v), v[len(v)//10], v[9*len(v)//10]))
PY
  rm -rf $O/tr_$f
  head -8 $O/joint32_one_stream_form_${f}_kernel_stats.csv | cut -c1-150
  cat $O/panel_by_grid_$f.txt
done
