#!/bin/bash
set -u
O=gpurun_out/r05_p2
mkdir -p $O
run() {  # name, env, args...
  local n=$1 e=$2; shift 2
  env $e timeout 300 python bench.py --no-cpu-baseline --merged-group 0 "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_$n.json") if l.startswith('{"metric"')][-1])
    print("$n", d["value"], d["ms_per_step"], "in flight", d["config"]["batches_in_flight"], "timeouts", d.get("lstm_handoff_timeouts"))
except Exception as e:
    print("$n failed", e); print(open("$O/bench_$n.err").read()[-1500:])
PY
}
run pipe2_share2 "APS_PIPE_SHARE=2" --pipeline 2
run pipe3_share2 "APS_PIPE_SHARE=2" --pipeline 3
run pipe2_share4 "APS_PIPE_SHARE=4" --pipeline 2
run pipe3_share4 "APS_PIPE_SHARE=4" --pipeline 3
run pipe3_share2_q8 "APS_PIPE_SHARE=2 GPU_MAX_HW_QUEUES=8" --pipeline 3
