#!/bin/bash
# Round 5: the whole GPU suite + smoke + the default bench line (no CPU-oracle leg) on the current tree.
set -u
O=gpurun_out/r05_check
mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/pytest_gpu_tail.txt
cat $O/pytest_gpu_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
tail -3 $O/smoke.txt
timeout 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
try:
    d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    r=d["roofline"]; sr=d["stage_roofline"]
    print(d["value"], d["ms_per_step"], "one-stream", d.get("single_stream_ms_per_step"), "default one-stream", d.get("single_stream_default_ms_per_step"), d.get("single_stream_default_value"), r["kernel"][:40], r["kernel_ms_per_step"], r["frac"], "stages", sr["all_stages"]["survey_8d"]["frac"], "merged", d.get("merged_batch", {}).get("value"))
except Exception as e:
    print("bench failed", e); print(open("$O/bench.err").read()[-2000:])
PY
