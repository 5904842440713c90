#!/bin/bash
# round 4, visit 2: cold weights and the next-image hint
set -u
O=gpurun_out/r04_s2
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest: GEMM kernels (panel) =="
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x --tb=short -k "(linear or fp16x2) and (panel or one-tile)" > $O/pytest_gemm.log 2>&1; tail -3 $O/pytest_gemm.log
echo "== shape probe with cold weights =="
timeout 600 python scripts/panel_gemm_probe.py 3 2>&1 | tee $O/probe_cold.txt
for pf in 1 0; do
echo "== joint bench group 1: panel, prefetch $pf =="
APS_GEMM_PREFETCH=$pf timeout 600 python bench.py --group 1 --steps 40 --warmup 5 --no-cpu-baseline 2> $O/bench_g1_pf$pf.err | tail -1 > $O/bench_g1_pf$pf.json
done
echo "== joint bench group 1: planes-pass form =="
APS_GEMM_SPLIT_LAYOUT=2 timeout 600 python bench.py --group 1 --steps 40 --warmup 5 --no-cpu-baseline 2> $O/bench_g1_l2.err | tail -1 > $O/bench_g1_l2.json
python - <<'PY'
import json
for n in ("pf1","pf0","l2"):
    try:
        d=json.load(open(f"gpurun_out/r04_s2/bench_g1_{n}.json"))
        print(n, "value", d["value"], "ms", d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), "gemm ms", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"], d["stage_us"])
    except Exception as e:
        print(n, "failed", e)
PY
