#!/bin/bash
set -u
O=gpurun_out/r05_p3
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
    print("$n failed", e); print(open("$O/bench_$n.err").read()[-600:])
PY
}
Q="GPU_MAX_HW_QUEUES=8"
run rep2_q8 "$Q" --replicas 2
run rep3_q8 "$Q" --replicas 3
run rep4_q8 "$Q" --replicas 4
run pipe3_s2_q8 "$Q APS_PIPE_SHARE=2" --pipeline 3
run pipe4_s2_q8 "$Q APS_PIPE_SHARE=2" --pipeline 4
run pipe5_s2_q8 "$Q APS_PIPE_SHARE=2" --pipeline 5
run pipe4_s1_q8 "$Q APS_PIPE_SHARE=1" --pipeline 4
run pipe4_s3_q8 "$Q APS_PIPE_SHARE=3" --pipeline 4
run pipe4_s2_q16 "GPU_MAX_HW_QUEUES=16 APS_PIPE_SHARE=2" --pipeline 4
