#!/bin/bash
# Round 6, visit 7: form f with the residual tile + next-image prefetch requested from inside the last chunk, DPP maxima
set -u
R=$(pwd)
O=gpurun_out/r06_s7
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -k "gemm or fp16x2 or linear or panel" > $O/pytest_gemm.log 2>&1
tail -3 $O/pytest_gemm.log
for f in e f; do
  APS_PANEL_FORM=$f timeout 600 python bench.py --no-cpu-baseline --merged-group 0 --no-host-input > $O/bench_$f.log 2>&1
  grep '^{"metric"' $O/bench_$f.log | tail -1 > $O/bench_$f.json
  python - <<PY
import json
d=json.load(open("$O/bench_$f.json"))
print("form $f:", d["value"], d["ms_per_step"], "single", d.get("single_stream_value"), "roof", d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("kernel_ms_per_step"))
PY
done
f=f
(cd /tmp && env APS_PANEL_FORM=$f APS_GEMM_KGROUP=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr_$f -o t -- \
   python $R/bench.py --group 1 --merged-group 0 --replicas 1 --pipeline 0 --steps 40 --warmup 5 --no-cpu-baseline > $R/$O/tr_$f.log 2>&1)
fcsv=$(find $O/tr_$f -name "*kernel_stats.csv" | head -1)
[ -n "$fcsv" ] && cp "$fcsv" $O/joint32_one_stream_form_${f}_kernel_stats.csv
rm -rf $O/tr_$f
head -12 $O/joint32_one_stream_form_${f}_kernel_stats.csv | cut -c1-130
