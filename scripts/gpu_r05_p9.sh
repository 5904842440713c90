#!/bin/bash
set -u
O=gpurun_out/r05_p9
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
run head3 A=1
run rotate4 APS_PIPE_FRONT=rotate APS_BENCH_PIPELINE=4
run rotate3 APS_PIPE_FRONT=rotate APS_BENCH_PIPELINE=3
run rotate5 APS_PIPE_FRONT=rotate APS_BENCH_PIPELINE=5
run rotate4_s1 APS_PIPE_FRONT=rotate APS_BENCH_PIPELINE=4 APS_PIPE_SHARE=1
