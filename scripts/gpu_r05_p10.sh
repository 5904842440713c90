#!/bin/bash
set -u
O=gpurun_out/r05_p10
mkdir -p $O
run() {
  tag=$1; shift
  env "$@" APS_BENCH_NO_HOST_INPUT=1 timeout 300 python bench.py --no-cpu-baseline --merged-group 0 > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"])
except Exception as e:
    print("$tag failed", e); print(open("$O/$tag.err").read()[-1500:])
PY
}
run head4 APS_BENCH_PIPELINE=4
run head4_pad APS_BENCH_PIPELINE=4 APS_PIPE_PAD=1
run head3_pad APS_PIPE_PAD=1
run head4_q5 APS_BENCH_PIPELINE=4 GPU_MAX_HW_QUEUES=5
run head4_q12 APS_BENCH_PIPELINE=4 GPU_MAX_HW_QUEUES=12
run head5 APS_BENCH_PIPELINE=5
run head6 APS_BENCH_PIPELINE=6
