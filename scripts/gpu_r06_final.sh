#!/bin/bash
# Last visits of round 6: the whole GPU suite, smoke, the driver-style bench lines (default flags, and a short --steps 20)
set -u
O=gpurun_out/r06_final
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
echo "== smoke =="
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1 | tee $O/smoke.log
echo "== bench (driver style) =="
timeout 600 python bench.py 2> $O/bench_joint.err | tail -1 > $O/bench_joint.json; cut -c1-220 $O/bench_joint.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_joint_s20.err | tail -1 > $O/bench_joint_s20.json
python - <<'PY'
import json
for f in ("bench_joint", "bench_joint_s20"):
    try:
        d=json.load(open(f"gpurun_out/r06_final/{f}.json"))
    except Exception as e:
        print(f, "failed", e); continue
    r=d["roofline"]; m=d.get("merged_batch") or {}
    print(f, "headline", d["value"], d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), "frac", r["frac"], "in flight", r.get("launches_in_flight"), "per launch", r.get("per_launch", {}).get("ms"), "traffic", r.get("traffic"))
    print("  host_input", (d.get("host_input") or {}).get("value"), "latency", {k: v for k, v in (d.get("latency_ms_per_batch") or {}).items() if k != "note"}, "stages", d.get("stage_ms_under_load"))
    if m: print("  merged", m["value"], m["ms_per_step"])
    print("  8d frac", d["stage_roofline"]["all_stages"]["survey_8d"]["frac"], "in flight", (d["stage_roofline"].get("in_flight") or {}).get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "parity", d.get("parity"))
PY
