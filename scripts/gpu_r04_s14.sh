#!/bin/bash
# round 4, visit 14: aps_gemm_tn (weight + bias gradients without transposed copies): parity, the training
# step, its kernel trace; the pruned panel forms
set -u
O=gpurun_out/r04_s14
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_train.txt
tail -5 $O/pytest_train.txt
timeout 600 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "linear or fp16x2 or gemm" 2>&1 | tail -5 > $O/pytest_gemm.txt
tail -3 $O/pytest_gemm.txt
timeout 600 python bench.py --workload train --no-cpu-baseline 2> $O/bench_train.err | tail -1 > $O/bench_train.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_s14/bench_train.json"))
print("train: value", d["value"], d["unit"], "ms/step", d["ms_per_step"])
PY
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_train -o train -- python /root/repo/bench.py --workload train --no-cpu-baseline > /root/repo/$O/prof_train.log 2>&1
cd /root/repo
f=$(find $O/prof_train -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/train_kernel_stats.csv && head -25 $O/train_kernel_stats.csv | cut -c1-160
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_default.err | tail -1 > $O/bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_s14/bench_default.json"))
print("default: value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "merged", d.get("merged_batch",{}).get("value"))
PY
