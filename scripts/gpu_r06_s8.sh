#!/bin/bash
set -u
O=gpurun_out/r06_s8
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -k "gemm or fp16x2 or linear or panel" > $O/pytest_gemm.log 2>&1
tail -3 $O/pytest_gemm.log
for f in e g h; do
  APS_PANEL_FORM=$f timeout 600 python bench.py --no-cpu-baseline --merged-group 0 --no-host-input > $O/bench_$f.log 2>&1
  grep '^{"metric"' $O/bench_$f.log | tail -1 > $O/bench_$f.json
  python - <<PY
import json
d=json.load(open("$O/bench_$f.json"))
print("form $f:", d["value"], d["ms_per_step"], "single", d.get("single_stream_value"), "roof", d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("kernel_ms_per_step"))
PY
done
