#!/bin/bash
# Round 5: the pipeline headline under other GEMM tile forms / LSTM shares (one box, back to back)
set -u
O=gpurun_out/r05_p5
mkdir -p $O
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
run base A=1
run form_c APS_PANEL_FORM=c
run form_a APS_PANEL_FORM=a
run share3 APS_PIPE_SHARE=3
run base2 A=1
