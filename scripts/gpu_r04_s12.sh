#!/bin/bash
set -u
O=gpurun_out/r04_s12
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q --tb=short -k "occ4" 2>&1 | tail -3
timeout 300 python scripts/panel_gemm_probe.py 2,31,35 2>&1 | grep -v amdgpu.ids | tee $O/probe.txt
for f in 1 5; do timeout 300 python scripts/stream_overlap_probe.py $f 2>&1 | grep -v amdgpu.ids | tee -a $O/stream_overlap.txt; done
for f in a e; do
APS_PANEL_FORM=$f timeout 600 python bench.py --group 1 --merged-group 0 --steps 40 --warmup 5 --no-cpu-baseline 2> $O/bench_g1_$f.err | tail -1 > $O/bench_g1_$f.json
done
python - <<'PY'
import json
for n in ("a","e"):
    try:
        d=json.load(open(f"gpurun_out/r04_s12/bench_g1_{n}.json"))
        print(n, "value", d["value"], "ms", d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), "gemm ms", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"])
    except Exception as e:
        print(n, "failed", e)
PY
