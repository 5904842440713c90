#!/bin/bash
# round 4, visit 6: panel kernel v2 (staging of the next chunk inside the MFMA loop, early epilogue operands)
set -u
O=gpurun_out/r04_s6
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest: GEMM kernels (panel forms) =="
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x --tb=short -k "(linear or fp16x2) and panel" > $O/pytest_gemm.log 2>&1; tail -3 $O/pytest_gemm.log
echo "== shape probe =="
timeout 600 python scripts/panel_gemm_probe.py 2,31,32,33,34 2>&1 | grep -v amdgpu.ids | tee $O/probe.txt
echo "== traces =="
for args in "2016 512 512 1" "2016 1024 512 1 ln" "2016 512 1024 1" "8064 1024 512 3 ln"; do
  APS_AMD_LIB=$PWD/aps_amd/csrc/libaps_amd_ptrace.so timeout 120 python scripts/panel_trace.py $args 2>&1 | grep -v amdgpu.ids | tee -a $O/trace.txt
done
echo "== joint bench group 1 =="
timeout 600 python bench.py --group 1 --steps 40 --warmup 5 --no-cpu-baseline 2> $O/bench_g1.err | tail -1 > $O/bench_g1.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_s6/bench_g1.json"))
print("value", d["value"], "ms", d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), "gemm ms", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"], d["stage_us"])
PY
