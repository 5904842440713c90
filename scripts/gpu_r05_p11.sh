#!/bin/bash
set -u
O=gpurun_out/r05_p11
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
run base A=1
run kgroup APS_EXPERIMENT_KGROUP_ALWAYS=1
run kgroup_w2 APS_EXPERIMENT_KGROUP_ALWAYS=1 APS_BENCH_PIPELINE=2
run base2 A=1
