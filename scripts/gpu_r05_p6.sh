#!/bin/bash
set -u
O=gpurun_out/r05_p6
mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --merged-group 0 > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], "one-stream", d.get("single_stream_ms_per_step"), "x2", d.get("whole_step_replicas", {}).get("value"))
except Exception as e:
    print("$tag failed", e); print(open("$O/$tag.err").read()[-1500:])
PY
}
run base A=1
run nolstm_w3 APS_EXPERIMENT_SKIP_LSTM=1
run nolstm_w2 APS_EXPERIMENT_SKIP_LSTM=1 APS_BENCH_PIPELINE=2
run nolstm_w4 APS_EXPERIMENT_SKIP_LSTM=1 APS_BENCH_PIPELINE=4
run nolstm_w6 APS_EXPERIMENT_SKIP_LSTM=1 APS_BENCH_PIPELINE=6
