#!/bin/bash
# round 4, visit 4: the panel forms (32/64 rows x 128/256 columns), late A requests
set -u
O=gpurun_out/r04_s4
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest: GEMM kernels (panel forms) =="
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x --tb=short -k "(linear or fp16x2) and panel" > $O/pytest_gemm.log 2>&1; tail -3 $O/pytest_gemm.log
echo "== shape probe =="
timeout 600 python scripts/panel_gemm_probe.py 2,31,32,33,34 2>&1 | tee $O/probe.txt
for f in a b; do
echo "== joint bench group 1, form $f =="
APS_PANEL_FORM=$f timeout 600 python bench.py --group 1 --steps 40 --warmup 5 --no-cpu-baseline 2> $O/bench_g1_$f.err | tail -1 > $O/bench_g1_$f.json
done
python - <<'PY'
import json
for n in ("a","b"):
    try:
        d=json.load(open(f"gpurun_out/r04_s4/bench_g1_{n}.json"))
        print(n, "value", d["value"], "ms", d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), "gemm ms", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"], d["stage_us"])
    except Exception as e:
        print(n, "failed", e)
PY
