#!/bin/bash
# round 4, visit 13: three batches in flight; group 1 with the shipped dispatch; merged batch
set -u
O=gpurun_out/r04_s13
mkdir -p $O
export TMPDIR=/tmp
for r in 2 3; do
timeout 600 python bench.py --group 1 --merged-group 0 --replicas $r --steps 60 --warmup 5 --no-cpu-baseline 2> $O/bench_g1_r$r.err | tail -1 > $O/bench_g1_r$r.json
done
timeout 600 python bench.py --group 4 --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_g4.err | tail -1 > $O/bench_g4.json
APS_PANEL_FORM=e timeout 600 python bench.py --group 4 --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_g4_e.err | tail -1 > $O/bench_g4_e.json
python - <<'PY'
import json
for n in ("g1_r2","g1_r3","g4","g4_e"):
    try:
        d=json.load(open(f"gpurun_out/r04_s13/bench_{n}.json"))
        print(n, "value", d["value"], "ms", d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), "gemm ms", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"].get("other_gemm_kernels"))
    except Exception as e:
        print(n, "failed", e)
PY
