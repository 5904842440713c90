# round 3, session 16: two K steps per barrier for the fp16 two-plane GEMM's launches with two workgroups per CU
O=gpurun_out/r03_s16
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_joint.py -q -m gpu -x -k "fp16x2 or linear or gemm or config4 or config5 or joint" > $O/pytest_gemm.log 2>&1
echo "gemm tests exit $?"; tail -3 $O/pytest_gemm.log | cut -c1-220
timeout 200 python scripts/joint_gemm_shapes.py 4 2>&1 | grep -E "GEMM calls|512, 512\)|512, 1024\)"
APS_GEMM_PAIR_TILES=0 timeout 200 python scripts/joint_gemm_shapes.py 4 2>&1 | grep -E "GEMM calls|512, 512\)|512, 1024\)"
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-baseline-batch > $O/joint_$tag.json 2> $O/joint_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/joint_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), "gemm ms", d["roofline"].get("kernel_ms_per_step"), "frac", d["roofline"]["frac"])
except Exception as e:
    print("$tag failed", e); print(open("$O/joint_$tag.err").read()[-1500:])
PY
}
run pair X=1
run off APS_GEMM_PAIR_TILES=0
run pair_again X=1
run off_again APS_GEMM_PAIR_TILES=0
