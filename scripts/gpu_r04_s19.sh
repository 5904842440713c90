#!/bin/bash
# round 4, visit 19: the GEMM timing of the bench line by back-to-back re-issue, beside the bracket figures
set -u
O=gpurun_out/r04_s19
mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_joint.err | tail -1 > $O/bench_joint.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_s19/bench_joint.json"))
for name, r in (("32", d["roofline"]), ("128", d["merged_batch"]["roofline"])):
    print(name, "value", d["value"] if name=="32" else d["merged_batch"]["value"], "frac", r["frac"], "kernel ms", r["kernel_ms_per_step"], "us/launch", r.get("kernel_us_per_launch"), "bracket", r["bracketed_ms_per_step"], r.get("bracket_corrected_ms_per_step"), r["timing"][:40], r.get("other_gemm_kernels"))
PY

