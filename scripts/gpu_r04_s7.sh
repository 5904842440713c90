#!/bin/bash
# round 4, visit 7: do two batches in flight overlap?  kernel trace of the 32-utterance step on two streams
set -u
O=gpurun_out/r04_s7
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest: GEMM kernels (panel forms) =="
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x --tb=short -k "(linear or fp16x2) and panel" > $O/pytest_gemm.log 2>&1; tail -3 $O/pytest_gemm.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_g1_r2 -o trace -- \
   python $R/bench.py --group 1 --replicas 2 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline > $R/$O/bench_g1_r2_prof.json 2> $R/$O/bench_g1_r2_prof.err)
python scripts/trace_overlap.py $(find $O/prof_g1_r2 -name "*kernel_trace.csv" | head -1) 0.6 | tee $O/overlap_g1_r2.txt
cp $(find $O/prof_g1_r2 -name "*kernel_stats.csv" | head -1) $O/g1_r2_kernel_stats.csv
rm -rf $O/prof_g1_r2
for f in a c; do
echo "== joint bench group 1, form $f =="
APS_PANEL_FORM=$f timeout 600 python bench.py --group 1 --steps 40 --warmup 5 --no-cpu-baseline 2> $O/bench_g1_$f.err | tail -1 > $O/bench_g1_$f.json
done
python - <<'PY'
import json
for n in ("a","c"):
    try:
        d=json.load(open(f"gpurun_out/r04_s7/bench_g1_{n}.json"))
        print(n, "value", d["value"], "ms", d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), "gemm ms", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"])
    except Exception as e:
        print(n, "failed", e)
PY
