#!/bin/bash
# round 4, visit 8: fused STFT + features (frame-major), covariance + solve in one launch
set -u
O=gpurun_out/r04_s8
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest: front end, MVDR, joint =="
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_joint.py tests/test_gpu_cplx.py tests/test_gpu_backward.py -m gpu -q -x --tb=short > $O/pytest_front.log 2>&1; tail -4 $O/pytest_front.log
echo "== frontend bench: default =="
timeout 300 python bench.py --workload frontend --no-cpu-baseline 2> $O/fe_default.err | tail -1 > $O/fe_default.json
echo "== frontend bench: two launches for STFT / features =="
timeout 300 python bench.py --workload frontend --no-cpu-baseline --no-fuse-features 2> $O/fe_nofuse.err | tail -1 > $O/fe_nofuse.json
echo "== frontend bench: round 3's MVDR sequence =="
APS_MVDR_FOUR_LAUNCHES=1 timeout 300 python bench.py --workload frontend --no-cpu-baseline 2> $O/fe_mvdr4.err | tail -1 > $O/fe_mvdr4.json
echo "== frontend bench: one stream =="
timeout 300 python bench.py --workload frontend --no-cpu-baseline --replicas 1 2> $O/fe_r1.err | tail -1 > $O/fe_r1.json
python - <<'PY'
import json
for n in ("default","nofuse","mvdr4","r1"):
    try:
        d=json.load(open(f"gpurun_out/r04_s8/fe_{n}.json"))
        sr=d["stage_roofline"]
        print(n, "value", d["value"], "ms", d["ms_per_step"], {k:(v["us_per_launch"], v["frac"]) for k,v in sr.items() if isinstance(v,dict) and "us_per_launch" in v}, "all", sr["all_stages"]["us_per_batch"], sr["all_stages"]["frac"], "8d", sr["all_stages"]["survey_8d"]["frac"])
    except Exception as e:
        print(n, "failed", e)
        import subprocess; print(subprocess.run(["tail","-5",f"gpurun_out/r04_s8/fe_{n}.err"],capture_output=True,text=True).stdout)
PY
echo "== joint bench (default: group 1 + merged) =="
timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench_joint.err | tail -1 > $O/bench_joint.json
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r04_s8/bench_joint.json"))
    m=d.get("merged_batch",{})
    print("value", d["value"], "ms", d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), d.get("single_stream_value"), "gemm", d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"], "parity", d.get("parity"))
    print("stage", {k:(v["us_per_launch"], v["frac"]) for k,v in d["stage_roofline"].items() if isinstance(v,dict) and "us_per_launch" in v}, d["stage_roofline"]["all_stages"])
    print("merged", m.get("value"), m.get("ms_per_step"), m.get("single_stream_ms_per_step"), m.get("roofline",{}).get("frac"), m.get("stage_roofline",{}).get("all_stages"))
    print("cpu", d.get("cpu_baseline",{}).get("value"))
except Exception as e:
    print("failed", e)
    import subprocess; print(subprocess.run(["tail","-8","gpurun_out/r04_s8/bench_joint.err"],capture_output=True,text=True).stdout)
PY
