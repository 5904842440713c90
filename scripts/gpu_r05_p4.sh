#!/bin/bash
set -u
O=gpurun_out/r05_p4
mkdir -p $O
run() {  # name, env, args...
  local n=$1 e=$2; shift 2
  env $e timeout 300 python bench.py --no-cpu-baseline --merged-group 0 "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_$n.json") if l.startswith('{"metric"')][-1])
    print("$n", d["value"], d["ms_per_step"], d["ms_per_step_regions"]["min"], d["ms_per_step_regions"]["max"], "timeouts", d.get("lstm_handoff_timeouts"))
except Exception as e:
    print("$n failed", e); print(open("$O/bench_$n.err").read()[-600:])
PY
}
B="GPU_MAX_HW_QUEUES=8 APS_PIPE_SHARE=2"
run base "$B" --pipeline 3
run base_again "$B" --pipeline 3
run prio "$B APS_PIPE_LSTM_PRIORITY=1" --pipeline 3
run q6 "GPU_MAX_HW_QUEUES=6 APS_PIPE_SHARE=2" --pipeline 3
run shape21 "$B APS_LSTM_SHAPE=2,1" --pipeline 3
run shape14 "$B APS_LSTM_SHAPE=1,4" --pipeline 3
run steps200 "$B" --pipeline 3 --steps 200
run rep2 "GPU_MAX_HW_QUEUES=8" --replicas 2
