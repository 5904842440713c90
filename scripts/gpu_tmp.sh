#!/bin/bash
set -u
O=gpurun_out/r06_m15
mkdir -p $O
run() {  # tag, env, args...
  local tag=$1 e=$2; shift 2
  env $e timeout 600 python bench.py --no-cpu-baseline --merged-group 0 --no-host-input "$@" > $O/bench_$tag.log 2>&1
  grep '^{"metric"' $O/bench_$tag.log | tail -1 > $O/bench_$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$tag.json"))
    print("$tag:", d["value"], d["ms_per_step"], "stages", d.get("stage_ms_under_load"), "timeouts", d.get("lstm_handoff_timeouts"))
except Exception as e:
    print("$tag: FAILED", e); import subprocess; print(subprocess.run(["tail","-3","$O/bench_$tag.log"],capture_output=True,text=True).stdout[-800:])
PY
}
run base APS_X=1
run s14 APS_LSTM_SHAPE=1,4
run s12 APS_LSTM_SHAPE=1,2
run s21 APS_LSTM_SHAPE=2,1
run s11 APS_LSTM_SHAPE=1,1
