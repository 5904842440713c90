#!/bin/bash
# Round 5, last visit: the front-end and one-stream lines from the final tree (ABI 56)
set -u
O=gpurun_out/r05_bench2
mkdir -p $O
timeout 60 python bench.py --workload frontend --no-cpu-baseline 2> $O/bench_frontend.err | tail -1 > $O/bench_frontend.json
timeout 60 python bench.py --group 1 --merged-group 0 --replicas 1 --steps 40 --warmup 5 --no-cpu-baseline 2> $O/bench_joint32_one_stream.err | tail -1 > $O/bench_joint32_one_stream.json
python - <<'PY'
import json
for n in ("frontend","joint32_one_stream"):
    try:
        d=json.load(open(f"gpurun_out/r05_bench2/bench_{n}.json"))
        print(n, d["value"], d["unit"], "ms", d["ms_per_step"])
    except Exception as e:
        print(n, "failed", e)
PY
