#!/bin/bash
# round 4, visit 17: the training step's projections on the two-plane GEMMs (per-step weight images)
set -u
O=gpurun_out/r04_s17
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "one_launch_weight_image or training_projections" 2>&1 | tail -15 > $O/pytest_image.txt; tail -4 $O/pytest_image.txt
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py tests/test_gpu_dccrn.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_train.txt; tail -4 $O/pytest_train.txt
timeout 600 python bench.py --workload train --no-cpu-baseline 2> $O/bench_train.err | tail -1 > $O/bench_train.json
APS_TRAIN_SPLIT=0 timeout 600 python bench.py --workload train --no-cpu-baseline 2> $O/bench_train_f32.err | tail -1 > $O/bench_train_f32.json
python - <<'PY'
import json
for n in ("train","train_f32"):
    d=json.load(open(f"gpurun_out/r04_s17/bench_{n}.json"))
    print(n, d["value"], d["unit"], "ms/step", d["ms_per_step"], d.get("loss_first_last"), d.get("fp32_path_tiles"))
PY
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -o t -- python /root/repo/bench.py --workload train --no-cpu-baseline > /root/repo/$O/prof.log 2>&1)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/train_kernel_stats.csv && head -14 $O/train_kernel_stats.csv | cut -c1-150
rm -rf $O/prof
