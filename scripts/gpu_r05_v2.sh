#!/bin/bash
# Round 5: the tests added since the last GPU visit, then the default bench line (pipeline headline) with and
# without the in-process GPU_MAX_HW_QUEUES default.
set -u
O=gpurun_out/r05_v2
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_replicas.py tests/test_gpu_multi.py -q -m gpu -x 2>&1 | tail -15 > $O/pytest_new.txt
timeout 300 python -m pytest tests/test_gpu_joint.py -q -m gpu -x -k "reuses or headline" 2>&1 | tail -8 >> $O/pytest_new.txt
cat $O/pytest_new.txt
show() {
python - <<PY
import json
try:
    d=json.loads(open("$1").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("$1", d["value"], d["ms_per_step"], "one-stream", d.get("single_stream_ms_per_step"), "whole-step x2", d.get("whole_step_replicas", {}).get("value"), r["kernel"][:30], r["kernel_ms_per_step"], r["frac"], "merged", d.get("merged_batch", {}).get("value"), d["config"].get("batches_in_flight"))
except Exception as e:
    print("bench failed", e); print(open("$2").read()[-2000:])
PY
}
timeout 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; show $O/bench.json $O/bench.err
GPU_MAX_HW_QUEUES=4 timeout 300 python bench.py --no-cpu-baseline --merged-group 0 > $O/bench_q4.json 2> $O/bench_q4.err; show $O/bench_q4.json $O/bench_q4.err
APS_BENCH_MERGED_PIPELINE=3 timeout 400 python bench.py --no-cpu-baseline > $O/bench_mp.json 2> $O/bench_mp.err; show $O/bench_mp.json $O/bench_mp.err
