#!/bin/bash
# round 4, visit 1: the panel GEMM -- parity of every GEMM form, per-shape times against the planes-pass
# form, the joint step with either form; the soffset range probe.
set -u
O=gpurun_out/r04_s1
mkdir -p $O
export TMPDIR=/tmp
echo "== soffset range probe =="
timeout 60 aps_amd/csrc/_micro/soffset_range 2>&1 | tee $O/soffset_range.txt
echo "== pytest: GEMM kernels =="
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x --tb=short -k "linear or fp16x2 or split_planes" > $O/pytest_gemm.log 2>&1; tail -5 $O/pytest_gemm.log
echo "== shape probe, auto panel height =="
timeout 300 python scripts/panel_gemm_probe.py 2,3 2>&1 | tee $O/probe_auto.txt
echo "== shape probe, 32-row panels =="
APS_PANEL_ROWS=32 timeout 300 python scripts/panel_gemm_probe.py 3 2>&1 | tee $O/probe_rt32.txt
echo "== shape probe, 64-row panels =="
APS_PANEL_ROWS=64 timeout 300 python scripts/panel_gemm_probe.py 3 2>&1 | tee $O/probe_rt64.txt
echo "== joint bench: panel (default) =="
timeout 600 python bench.py --steps 20 --warmup 5 2> $O/bench_joint_panel.err | tail -1 > $O/bench_joint_panel.json; cut -c1-400 $O/bench_joint_panel.json
echo "== joint bench: planes-pass form =="
APS_GEMM_SPLIT_LAYOUT=2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_joint_l2.err | tail -1 > $O/bench_joint_l2.json; cut -c1-400 $O/bench_joint_l2.json
python - <<'PY'
import json
for n in ("panel","l2"):
    try:
        d=json.load(open(f"gpurun_out/r04_s1/bench_joint_{n}.json"))
        b=d.get("baseline_batch",{})
        print(n, "value", d["value"], "ms", d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), "gemm ms", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"],
              "| b32", b.get("value"), b.get("ms_per_step"), "single", b.get("single_stream_ms_per_step"), "gemm", b.get("roofline",{}).get("kernel_ms_per_step"), "parity", d.get("parity"))
    except Exception as e:
        print(n, "failed", e)
PY
