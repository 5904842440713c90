#!/bin/bash
set -u
O=gpurun_out/r05_p1
mkdir -p $O
run() {  # name, args...
  local n=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --merged-group 0 "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_$n.json") if l.startswith('{"metric"')][-1])
    print("$n", d["value"], d["ms_per_step"], d["ms_per_step_regions"], "in flight", d["config"]["batches_in_flight"], "timeouts", d.get("lstm_handoff_timeouts"), d["launch"][:60])
except Exception as e:
    print("$n failed", e); print(open("$O/bench_$n.err").read()[-1500:])
PY
}
run default
run pipe2 --pipeline 2
run pipe3 --pipeline 3
run pipe4 --pipeline 4
