#!/bin/bash
# Round 6, visit 3: with the LDS-DMA panel form the workers' stage B shrinks -- is the head stream (A + LSTM + M) the
# bound now?  Stage times under load for forms e / f, bench lines with stage A / M on the workers.
set -u
O=gpurun_out/r06_s2
mkdir -p $O
for f in e f; do
  APS_PANEL_FORM=$f timeout 300 python scripts/pipeline_stage_times.py 3 2 head > $O/stages_${f}_head.txt 2>&1
  tail -8 $O/stages_${f}_head.txt
  APS_PANEL_FORM=$f timeout 300 python scripts/pipeline_stage_times.py 3 2 worker > $O/stages_${f}_worker.txt 2>&1
  tail -8 $O/stages_${f}_worker.txt
done
for cfg in "f head worker" "f worker worker" "e worker worker" "f worker head"; do
  set -- $cfg
  APS_PANEL_FORM=$1 timeout 600 python bench.py --no-cpu-baseline --merged-group 0 --no-host-input --pipe-front $2 --pipe-mid $3 > $O/bench_$1_$2_$3.log 2>&1
  grep '^{"metric"' $O/bench_$1_$2_$3.log | tail -1 > $O/bench_$1_$2_$3.json
  python - <<PY
import json
d=json.load(open("$O/bench_$1_$2_$3.json"))
print("form $1 front $2 mid $3:", d["value"], d["ms_per_step"], "single", d.get("single_stream_value"), "lat", d.get("latency_ms_per_batch",{}).get("headline"), "inflight", d.get("stage_roofline",{}).get("in_flight",{}).get("frac"))
PY
done
for w in 4 5; do
  APS_PANEL_FORM=f timeout 600 python bench.py --no-cpu-baseline --merged-group 0 --no-host-input --pipeline $w --pipe-front worker --pipe-mid worker > $O/bench_f_w$w.log 2>&1
  grep '^{"metric"' $O/bench_f_w$w.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f workers $w front/mid worker:', d['value'], d['ms_per_step'])"
done
timeout 600 python -m pytest tests/test_gpu_replicas.py -x -q -k "hardware_queues" -s 2>&1 | tail -5
