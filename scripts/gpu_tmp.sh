#!/bin/bash
O=gpurun_out/r06_covsw
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_joint.py -x -q 2>&1 | tail -3
for lib in libaps_amd.so libaps_amd_covnosw.so libaps_amd.so libaps_amd_covnosw.so; do
APS_AMD_LIB=$PWD/aps_amd/csrc/$lib timeout 600 python bench.py --workload frontend --no-cpu-baseline > $O/bench_frontend_$lib.log 2>&1
grep '^{"metric"' $O/bench_frontend_$lib.log | tail -1 > $O/bench_frontend_$lib.json
python - <<PY
import json
d=json.load(open("$O/bench_frontend_$lib.json"))
print("$lib frontend:", d["value"], d["ms_per_step"], {k:(v.get("us_per_launch"), v.get("frac")) for k,v in d["stage_roofline"].items() if isinstance(v,dict) and "frac" in v})
PY
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_s20.log 2>&1
grep '^{"metric"' $O/bench_s20.log | tail -1 > $O/bench_s20.json
python - <<PY
import json
d=json.load(open("$O/bench_s20.json"))
print("s20:", d["value"], d["ms_per_step"], "steady", d.get("steady_state"), "traffic", d["roofline"].get("traffic"), "host", (d.get("host_input") or {}).get("value"))
PY
