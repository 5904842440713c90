#!/bin/bash
# Round 6: every bench line from the shipped tree (driver-style default first)
set -u
O=gpurun_out/r06_bench
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 2> $O/bench_joint.err | tail -1 > $O/bench_joint.json
timeout 600 python bench.py --workload frontend 2> $O/bench_frontend.err | tail -1 > $O/bench_frontend.json
timeout 600 python bench.py --workload encoder 2> $O/bench_encoder.err | tail -1 > $O/bench_encoder.json
timeout 600 python bench.py --workload dccrn 2> $O/bench_dccrn.err | tail -1 > $O/bench_dccrn.json
timeout 600 python bench.py --workload train --no-cpu-baseline 2> $O/bench_train.err | tail -1 > $O/bench_train.json
timeout 600 python bench.py --group 1 --merged-group 0 --replicas 1 --steps 40 --warmup 5 --no-cpu-baseline 2> $O/bench_joint32_one_stream.err | tail -1 > $O/bench_joint32_one_stream.json
python - <<'PY'
import json
for n in ("joint","frontend","encoder","dccrn","train","joint32_one_stream"):
    try:
        d=json.load(open(f"gpurun_out/r06_bench/bench_{n}.json"))
        r=d.get("roofline") or {}
        print(n, d["value"], d["unit"], "ms", d["ms_per_step"], "frac", r.get("frac"), "traffic", r.get("traffic"), "merged", (d.get("merged_batch") or {}).get("value"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "failed", e)
PY
