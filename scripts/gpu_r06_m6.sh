#!/bin/bash
set -u
O=gpurun_out/r06_m6
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_mega.py tests/test_gpu_joint.py tests/test_gpu_replicas.py -x -q 2>&1 | tail -15
timeout 900 python bench.py > $O/bench_default.log 2>&1
grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
r=d["roofline"]
print("default:", d["value"], d["ms_per_step"], "single", d.get("single_stream_value"), "lat", d.get("latency_ms_per_batch"), "parity", d.get("parity"))
print("roofline:", r["frac"], r["achieved"], r["launches_in_flight"], r["per_launch"])
print("host_input:", d.get("host_input"))
print("in_flight:", d["stage_roofline"].get("in_flight"))
print("cpu:", d.get("cpu_baseline"))
print("timeouts", d.get("lstm_handoff_timeouts"), "merged", d.get("merged_batch",{}).get("value"))
PY
tail -3 $O/bench_default.log | cut -c1-400
