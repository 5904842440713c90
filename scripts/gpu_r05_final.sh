#!/bin/bash
# Last visit of round 5: the whole GPU suite, smoke, the driver-style bench line
set -u
O=gpurun_out/r05_final
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
echo "== smoke =="
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1 | tee $O/smoke.log
echo "== bench (driver style) =="
timeout 600 python bench.py --steps 20 --warmup 5 2> $O/bench_joint.err | tail -1 > $O/bench_joint.json; cut -c1-220 $O/bench_joint.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_final/bench_joint.json"))
r=d["roofline"]; m=d["merged_batch"]
print("headline", d["value"], d["ms_per_step"], "single", d["single_stream_ms_per_step"], "frac", r["frac"], "kernel ms", r["kernel_ms_per_step"], "us/launch", r["kernel_us_per_launch"], "traffic", r["traffic"])
print("merged", m["value"], m["ms_per_step"], "frac", m["roofline"]["frac"], m["roofline"]["kernel_ms_per_step"], m["roofline"].get("other_gemm_kernels"))
print("8d frac", d["stage_roofline"]["all_stages"]["survey_8d"]["frac"], "merged 8d", m["stage_roofline"]["all_stages"]["survey_8d"]["frac"], "cpu", d["cpu_baseline"]["value"], "parity", d["parity"])
PY
