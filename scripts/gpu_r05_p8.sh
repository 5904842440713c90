#!/bin/bash
set -u
O=gpurun_out/r05_p8
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_replicas.py -q -m gpu -x 2>&1 | tail -5
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --merged-group 0 > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], "one-stream", d.get("single_stream_ms_per_step"), "x2", d.get("whole_step_replicas", {}).get("value"))
except Exception as e:
    print("$tag failed", e); print(open("$O/$tag.err").read()[-1500:])
PY
}
run head A=1
run head_s1 APS_PIPE_SHARE=1
run head_s3 APS_PIPE_SHARE=3
run worker APS_PIPE_FRONT=worker
env timeout 300 python bench.py --no-cpu-baseline --merged-group 0 --batches 24 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"b24\", d[\"value\"], d[\"ms_per_step\"])"
