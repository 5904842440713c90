#!/bin/bash
set -u
O=gpurun_out/r06_m9
mkdir -p $O
run() {  # tag, env, args...
  local tag=$1 e=$2; shift 2
  env $e timeout 600 python bench.py --no-cpu-baseline --merged-group 0 --no-host-input "$@" > $O/bench_$tag.log 2>&1
  grep '^{"metric"' $O/bench_$tag.log | tail -1 > $O/bench_$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$tag.json"))
    print("$tag:", d["value"], d["ms_per_step"], "single", d.get("single_stream_value"), "lat", d.get("latency_ms_per_batch",{}).get("headline"), "timeouts", d.get("lstm_handoff_timeouts"))
except Exception as e:
    print("$tag: FAILED", e); import subprocess; print(subprocess.run(["tail","-5","$O/bench_$tag.log"],capture_output=True,text=True).stdout[-1500:])
PY
}
run default APS_X=1
run w5 APS_X=1 --pipeline 5
run w6_s1 APS_X=1 --pipe-share 1
run w6_s3 APS_X=1 --pipe-share 3
run w6_b18 APS_X=1 --batches 18
run w6_again APS_X=1
run w4 APS_X=1 --pipeline 4
