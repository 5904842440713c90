#!/bin/bash
set -u
O=gpurun_out/r05_p12
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_replicas.py tests/test_gpu_joint.py -q -m gpu -x -k "pipelined or staged" 2>&1 | tail -3
run() {
  tag=$1; shift
  env "$@" APS_BENCH_NO_HOST_INPUT=1 timeout 300 python bench.py --no-cpu-baseline --merged-group 0 > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], d["launch"][:90])
except Exception as e:
    print("$tag failed", e); print(open("$O/$tag.err").read()[-1500:])
PY
}
run base A=1
run mid_head APS_PIPE_MID=head
run base2 A=1
run mid_head2 APS_PIPE_MID=head
