#!/bin/bash
# round 4, visit 28: the MVDR adjoint's frame reductions as workgroup kernels: parity (both forms), training step
set -u
O=gpurun_out/r04_s28; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py tests/test_gpu_tasks.py -x -q -m gpu 2>&1 | tail -4 > $O/pytest_fast.txt; tail -2 $O/pytest_fast.txt
APS_GRAD_FUNCTORS=1 timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "mvdr or joint" 2>&1 | tail -2
timeout 600 python bench.py --workload train --no-cpu-baseline 2> $O/bench_train.err | tail -1 > $O/bench_train.json
APS_GRAD_FUNCTORS=1 timeout 600 python bench.py --workload train --no-cpu-baseline 2> $O/bench_train_f.err | tail -1 > $O/bench_train_functors.json
python - <<'PY'
import json
for n in ("train","train_functors"):
    d=json.load(open(f"gpurun_out/r04_s28/bench_{n}.json"))
    print(n, d["value"], d["unit"], "ms/step", d["ms_per_step"], d.get("loss_first_last"))
PY
