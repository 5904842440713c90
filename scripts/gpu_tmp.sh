#!/bin/bash
O=gpurun_out/r06_partition
mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-host-input ${EXTRA} > $O/bench_$tag.log 2>&1
  grep '^{"metric"' $O/bench_$tag.log | tail -1 > $O/bench_$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$tag.json"))
    print("$tag:", d["value"], d["ms_per_step"], d.get("stage_ms_under_load"), "timeouts", d.get("lstm_handoff_timeouts"))
except Exception as e:
    print("$tag: failed", e)
PY
  tail -2 $O/bench_$tag.log | grep -v '^{"metric"' | cut -c1-300
}
EXTRA="" run base A=1
EXTRA="--pipe-partition 2" run p2 A=1
EXTRA="--pipe-partition 2 --pipe-share 1" run p2_share1 A=1
EXTRA="--pipe-partition 3" run p3 A=1
EXTRA="--pipe-partition 4" run p4 A=1
EXTRA="--pipe-partition 2 --pipeline 7" run p2_w7 A=1
