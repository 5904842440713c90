#!/bin/bash
# round 4, visit 30: beamform + ASR features in one pass (8(d) P3): parity, the bench line with its stage table
set -u
O=gpurun_out/r04_s30; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_joint.py tests/test_gpu_parity.py tests/test_gpu_replicas.py -x -q -m gpu 2>&1 | tail -6 > $O/pytest.txt; tail -4 $O/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_joint.err | tail -1 > $O/bench_joint.json
APS_NO_BEAM_FEATURES=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_joint_two.err | tail -1 > $O/bench_joint_two_launches.json
python - <<'PY'
import json
for n in ("joint","joint_two_launches"):
    d=json.load(open(f"gpurun_out/r04_s30/bench_{n}.json"))
    sr=d["stage_roofline"]
    print(n, d["value"], d["ms_per_step"], "single", d["single_stream_ms_per_step"], {k:v["us_per_launch"] for k,v in sr.items() if isinstance(v,dict) and "us_per_launch" in v}, "8d", sr["all_stages"]["survey_8d"]["frac"], sr["all_stages"]["survey_8d"]["bytes_per_utterance"], "merged", d["merged_batch"]["value"], d["merged_batch"]["stage_roofline"]["all_stages"]["survey_8d"]["frac"], d.get("parity"))
PY
tail -3 $O/bench_joint.err
