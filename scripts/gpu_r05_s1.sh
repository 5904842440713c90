#!/bin/bash
# Round 5, visit 1: the K-group GEMM forms -- parity, per-shape timing against the four-wave form, the
# headline-batch oracle test, the stock torch launches left in the step, and the joint bench A/B.
set -u
O=gpurun_out/r05_s1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "linear_kernel or fp16x2 or layernorm_fold" 2>&1 | tail -5 > $O/pytest_gemm.txt
cat $O/pytest_gemm.txt
timeout 600 python -m pytest tests/test_gpu_joint.py -x -q -m gpu 2>&1 | tail -8 > $O/pytest_joint.txt
cat $O/pytest_joint.txt
timeout 600 python scripts/panel_gemm_probe.py 33,34,35 > $O/panel_probe.txt 2>&1
APS_KGROUP_RING=2 timeout 600 python scripts/panel_gemm_probe.py 34 > $O/panel_probe_ring2.txt 2>&1
cat $O/panel_probe.txt $O/panel_probe_ring2.txt
timeout 300 python scripts/step_torch_ops.py > $O/torch_ops.txt 2>&1
tail -40 $O/torch_ops.txt
timeout 600 python bench.py --no-cpu-baseline --merged-group 0 > $O/bench_k.json 2> $O/bench_k.err
APS_PANEL_FORM=e timeout 600 python bench.py --no-cpu-baseline --merged-group 0 > $O/bench_e.json 2> $O/bench_e.err
for f in k e; do python - <<PY
import json
d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$f", d["value"], d["ms_per_step"], "one-stream", d.get("single_stream_ms_per_step"), r["kernel"][:40], r["kernel_ms_per_step"], r["frac"], d.get("stage_us"))
PY
done
