#!/bin/bash
# Round 6, visit 6: PipelinedReplicas(lookahead): the head stream carries the LSTM launches only
set -u
O=gpurun_out/r06_s5
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_replicas.py -x -q 2>&1 | tail -3
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
    print("$tag: FAILED", e)
PY
}
run base_e APS_PANEL_FORM=e
run look_e_w3 APS_PANEL_FORM=e --pipe-front worker --pipe-mid worker --pipe-lookahead 1
run look_f_w3 APS_PANEL_FORM=f --pipe-front worker --pipe-mid worker --pipe-lookahead 1
run look_f_w4 APS_PANEL_FORM=f --pipeline 4 --pipe-front worker --pipe-mid worker --pipe-lookahead 1
run look_f_w3_s1 APS_PANEL_FORM=f --pipe-front worker --pipe-mid worker --pipe-lookahead 1 --pipe-share 1
run look_c_w3 APS_PANEL_FORM=c --pipe-front worker --pipe-mid worker --pipe-lookahead 1
run look_f_w3_midhead APS_PANEL_FORM=f --pipe-front worker --pipe-mid head --pipe-lookahead 1
tail -3 $O/bench_look_f_w3.log | cut -c1-300
