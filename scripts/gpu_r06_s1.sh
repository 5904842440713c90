#!/bin/bash
# Round 6, visit 2: the panel GEMM with the branch-free last chunk (all forms) and the LDS-DMA form 'f' --
# parity tests of every form, A/B bench lines, the workgroup trace of 'f' under load.
set -u
R=$(pwd)
O=gpurun_out/r06_s1
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -k "gemm or fp16x2 or linear or panel" > $O/pytest_gemm.log 2>&1
tail -3 $O/pytest_gemm.log
timeout 900 python -m pytest tests/test_gpu_replicas.py tests/test_gpu_joint.py -x -q > $O/pytest_joint.log 2>&1
tail -3 $O/pytest_joint.log
for f in e f; do
  APS_PANEL_FORM=$f timeout 600 python bench.py --no-cpu-baseline --merged-group 0 > $O/bench_$f.log 2>&1
  grep '^{"metric"' $O/bench_$f.log | tail -1 > $O/bench_$f.json
  python - <<PY
import json
d=json.load(open("$O/bench_$f.json"))
print("form $f:", d["value"], d["ms_per_step"], "single", d.get("single_stream_value"), "roof", d.get("roofline",{}).get("frac"))
PY
done
APS_PANEL_FORM=f APS_AMD_LIB=$R/aps_amd/csrc/libaps_amd_ptrace.so timeout 600 python scripts/panel_trace_under_load.py > $O/panel_trace_under_load_f.txt 2>&1
grep -A7 "N=512 K=512.*pipeline" $O/panel_trace_under_load_f.txt
APS_PANEL_FORM=e APS_AMD_LIB=$R/aps_amd/csrc/libaps_amd_ptrace.so timeout 600 python scripts/panel_trace_under_load.py > $O/panel_trace_under_load_e.txt 2>&1
grep -A7 "N=512 K=512.*pipeline" $O/panel_trace_under_load_e.txt
