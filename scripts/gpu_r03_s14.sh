# round 3, session 14: the 64 x 64 form of the fp16 two-plane GEMM for launches with few tiles
O=gpurun_out/r03_s14
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_joint.py tests/test_gpu_parity.py tests/test_gpu_decoder.py tests/test_gpu_dccrn.py -q -m gpu > $O/pytest_gemm.log 2>&1
echo "tests exit $?"; tail -3 $O/pytest_gemm.log | cut -c1-220
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline $EXTRA > $O/joint_$tag.json 2> $O/joint_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/joint_$tag.json").read().strip().splitlines()[-1])
    b=d.get("baseline_batch") or {}
    print("$tag", d["value"], d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), "gemm ms", d["roofline"].get("kernel_ms_per_step"), "frac", d["roofline"]["frac"], "| baseline", b.get("value"), b.get("ms_per_step"), (b.get("roofline") or {}).get("frac"))
except Exception as e:
    print("$tag failed", e); print(open("$O/joint_$tag.err").read()[-1500:])
PY
}
EXTRA=""
run new X=1
run old APS_GEMM_NARROW_TILES=0 APS_GEMM_SPLIT_MIN_TILES=256
run new_again X=1
EXTRA="--workload encoder"
run enc_new X=1
run enc_old APS_GEMM_NARROW_TILES=0 APS_GEMM_SPLIT_MIN_TILES=256
EXTRA="--workload dccrn"
run dccrn_new X=1
run dccrn_old APS_GEMM_NARROW_TILES=0 APS_GEMM_SPLIT_MIN_TILES=256
