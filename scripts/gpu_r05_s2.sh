#!/bin/bash
# Round 5, visit 2: chained projections (aps_linear_chain) and the one-launch MVDR tail -- parity, then A/B benches.
set -u
O=gpurun_out/r05_s2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "chain" 2>&1 | tail -15 > $O/pytest_chain.txt
cat $O/pytest_chain.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mvdr" 2>&1 | tail -8 > $O/pytest_mvdr.txt
cat $O/pytest_mvdr.txt
timeout 900 python -m pytest tests/test_gpu_joint.py tests/test_gpu_encoder.py -x -q -m gpu 2>&1 | tail -8 > $O/pytest_joint_encoder.txt
cat $O/pytest_joint_encoder.txt
run() {  # name, env...
  local n=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --merged-group 0 > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$n.json").read().strip().splitlines()[-1])
    r=d["roofline"]; sr=d["stage_roofline"]
    print("$n", d["value"], d["ms_per_step"], "one-stream", d.get("single_stream_ms_per_step"), r["kernel"][:48], r["kernel_ms_per_step"], r["frac"],
          "other", r.get("other_gemm_kernels"), "all", r.get("all_two_plane_gemms"), "mvdr_w us", sr["mvdr_weights"]["us_per_launch"], "stages frac", sr["all_stages"]["survey_8d"]["frac"])
except Exception as e:
    print("$n failed", e); print(open("$O/bench_$n.err").read()[-1500:])
PY
}
run chain_tail APS_X=1
run nochain_tail APS_GEMM_CHAIN=0
run chain_notail APS_MVDR_TAIL=0
run chain_wgs1024 APS_GEMM_CHAIN_WGS=1024
