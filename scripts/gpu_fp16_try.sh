# the two-plane fp16 GEMM (APS_GEMM_SPLIT_LAYOUT=2): parity tests, per-shape timing against the bf16
# form and the fp32 MFMA, per-kernel durations (row exponent pass / GEMM), the joint step with it
mkdir -p gpurun_out/r02_fp16
O=gpurun_out/r02_fp16
timeout 400 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "fp16" > $O/tests.log 2>&1
echo "tests exit $?"; tail -4 $O/tests.log
SPLIT_BENCH_ONLY=fp32,bd,fp16 timeout 300 python scripts/split_gemm_bench.py 8064 2016 > $O/gemm_shapes.txt 2> $O/gemm_shapes.err
cat $O/gemm_shapes.txt
( cd /tmp && export TMPDIR=/tmp && SPLIT_BENCH_ONLY=fp16 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o fp16 -- python $GRAFT_REPO_ROOT/scripts/split_gemm_bench.py 8064 > /dev/null 2>&1 )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200
APS_GEMM_SPLIT_LAYOUT=2 timeout 500 python bench.py > $O/joint_fp16.json 2> $O/joint_fp16.err
echo "fp16 bench exit $?"; tail -3 $O/joint_fp16.err
timeout 300 python bench.py --no-cpu-baseline > $O/joint_bd.json 2> $O/joint_bd.err
python - <<PY
import json
for tag in ("fp16", "bd"):
    try:
        d = json.loads(open("$O/joint_%s.json" % tag).read().strip().splitlines()[-1])
        print(tag, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"],
              d["roofline"].get("pipe", {}).get("frac"), d.get("parity"), d.get("single_stream_ms_per_step"))
    except Exception as e:
        print(tag, "failed", e)
PY
