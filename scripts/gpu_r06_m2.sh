#!/bin/bash
# Round 6: the joint bench with the conformer stack as one launch per batch (aps_amd.mega), workers = batches in flight
set -u
O=gpurun_out/r06_m2
mkdir -p $O
run() {  # tag, env, args...
  local tag=$1 e=$2; shift 2
  env $e timeout 600 python bench.py --no-cpu-baseline --merged-group 0 --no-host-input "$@" > $O/bench_$tag.log 2>&1
  grep '^{"metric"' $O/bench_$tag.log | tail -1 > $O/bench_$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$tag.json"))
    print("$tag:", d["value"], d["ms_per_step"], "single", d.get("single_stream_value"), "lat", d.get("latency_ms_per_batch",{}).get("headline"), "parity", d.get("parity"))
except Exception as e:
    print("$tag: FAILED", e); import subprocess; print(subprocess.run(["tail","-5","$O/bench_$tag.log"],capture_output=True,text=True).stdout[-1500:])
PY
}
run w3 APS_X=1 --pipeline 3
run w4 APS_X=1 --pipeline 4
run w5 APS_X=1 --pipeline 5
run w6 APS_X=1 --pipeline 6
run w7 GPU_MAX_HW_QUEUES=16 --pipeline 7
run w5_worker APS_X=1 --pipeline 5 --pipe-front worker --pipe-mid worker
run w6_q16 GPU_MAX_HW_QUEUES=16 --pipeline 6
