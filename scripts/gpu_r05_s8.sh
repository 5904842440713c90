#!/bin/bash
set -u
O=gpurun_out/r05_s8
mkdir -p $O
run() {  # name, env...
  local n=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --merged-group 0 > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$n.json").read().strip().splitlines()[-1])
    print("$n", d["value"], d["ms_per_step"], "one-stream", d.get("single_stream_ms_per_step"), "default", d.get("single_stream_default_ms_per_step"), "timeouts", d.get("lstm_handoff_timeouts"))
except Exception as e:
    print("$n failed", e); print(open("$O/bench_$n.err").read()[-800:])
PY
}
run default APS_X=1
run lstm_2_1 APS_LSTM_SHAPE=2,1
run lstm_1_1 APS_LSTM_SHAPE=1,1
run lstm_2_2 APS_LSTM_SHAPE=2,2
