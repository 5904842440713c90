#!/bin/bash
set -u
O=gpurun_out/r06_m5
mkdir -p $O
run() {  # tag, env, args...
  local tag=$1 e=$2; shift 2
  env $e timeout 600 python bench.py --no-cpu-baseline --merged-group 0 --no-host-input "$@" > $O/bench_$tag.log 2>&1
  grep '^{"metric"' $O/bench_$tag.log | tail -1 > $O/bench_$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$tag.json"))
    print("$tag:", d["value"], d["ms_per_step"], "single", d.get("single_stream_value"), "lat", d.get("latency_ms_per_batch",{}).get("headline"))
except Exception as e:
    print("$tag: FAILED", e); import subprocess; print(subprocess.run(["tail","-5","$O/bench_$tag.log"],capture_output=True,text=True).stdout[-1500:])
PY
}
run w4 APS_X=1 --pipeline 4
run w4_look APS_X=1 --pipeline 4 --pipe-front worker --pipe-mid worker --pipe-lookahead 1
run w5_look APS_X=1 --pipeline 5 --pipe-front worker --pipe-mid worker --pipe-lookahead 1
run w6_look APS_X=1 --pipeline 6 --pipe-front worker --pipe-mid worker --pipe-lookahead 1
run w6_look_midhead APS_X=1 --pipeline 6 --pipe-front worker --pipe-mid head --pipe-lookahead 1
run w4_s1 APS_X=1 --pipeline 4 --pipe-share 1
run w6_fronthead_look APS_X=1 --pipeline 6 --pipe-front head --pipe-mid worker --pipe-lookahead 1
