#!/bin/bash
# round 4, visit 9: covariance + solve with 16-bin workgroups; per-kernel durations of the front end; frames per wavefront
set -u
O=gpurun_out/r04_s9
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest: MVDR =="
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -k "mvdr or config2 or enh" > $O/pytest_mvdr.log 2>&1; tail -2 $O/pytest_mvdr.log
run() {  # name, env..., args
  local name=$1; shift
  env "$@" timeout 300 python bench.py --workload frontend --no-cpu-baseline --replicas 1 --steps 100 2> $O/fe_$name.err | tail -1 > $O/fe_$name.json
}
run default A=1
run mvdr4 APS_MVDR_FOUR_LAUNCHES=1
for it in 2 4 6; do run iters$it APS_STFT_ITERS=$it; done
python - <<'PY'
import json
for n in ("default","mvdr4","iters2","iters4","iters6"):
    try:
        d=json.load(open(f"gpurun_out/r04_s9/fe_{n}.json"))
        sr=d["stage_roofline"]
        print(n, "value", d["value"], {k:v["us_per_launch"] for k,v in sr.items() if isinstance(v,dict) and "us_per_launch" in v}, "all", sr["all_stages"]["us_per_batch"], "8d", sr["all_stages"]["survey_8d"]["frac"])
    except Exception as e:
        print(n, "failed", e)
PY
echo "== kernel trace, front end, one stream =="
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_fe -o trace -- \
   python $R/bench.py --workload frontend --replicas 1 --steps 100 --no-cpu-baseline > $R/$O/fe_prof.json 2> $R/$O/fe_prof.err)
f=$(find $O/prof_fe -name "*kernel_stats.csv" | head -1); cp $f $O/fe_r1_kernel_stats.csv; rm -rf $O/prof_fe
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r04_s9/fe_r1_kernel_stats.csv")))
for r in rows[:12]:
    print(f'{r["Name"][:80]:80s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:7.1f}')
PY
