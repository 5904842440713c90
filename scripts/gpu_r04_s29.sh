#!/bin/bash
# round 4, visit 29: channel-fastest GLU/dwconv weight gradient, one-launch column reductions: parity, training step
set -u
O=gpurun_out/r04_s29; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py tests/test_gpu_dccrn_train.py tests/test_gpu_tasks.py -x -q -m gpu 2>&1 | tail -4 > $O/pytest.txt; tail -2 $O/pytest.txt
timeout 600 python bench.py --workload train --no-cpu-baseline 2> $O/bench_train.err | tail -1 > $O/bench_train.json
APS_GRAD_FUNCTORS=1 timeout 600 python bench.py --workload train --no-cpu-baseline 2> $O/bench_train_f.err | tail -1 > $O/bench_train_functors.json
python - <<'PY'
import json
for n in ("train","train_functors"):
    d=json.load(open(f"gpurun_out/r04_s29/bench_{n}.json"))
    print(n, d["value"], d["unit"], "ms/step", d["ms_per_step"], d.get("loss_first_last"))
PY
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -o t -- python /root/repo/bench.py --workload train --no-cpu-baseline > /root/repo/$O/prof.log 2>&1)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/train_kernel_stats.csv
rm -rf $O/prof
