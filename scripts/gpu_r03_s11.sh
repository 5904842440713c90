# round 3, session 11: the GEMM's prologue (requests in one flight) and epilogue (row-major through LDS)
O=gpurun_out/r03_s11
mkdir -p $O
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/aps_amd/csrc
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_joint.py tests/test_gpu_parity.py -q -m gpu -x -k "fp16x2 or linear or gemm or config4 or config5 or joint or encoder" > $O/pytest_gemm.log 2>&1
echo "gemm tests exit $?"; tail -6 $O/pytest_gemm.log | cut -c1-220
APS_AMD_LIB=$L/libaps_amd_trace.so timeout 200 python scripts/gemm_trace.py 8064 1024 512 ln 2>&1 | grep -E "per call|1008 workgroups|prologue"
APS_AMD_LIB=$L/libaps_amd_trace.so timeout 200 python scripts/gemm_trace.py 8064 512 512 2>&1 | grep -E "per call|workgroups on|prologue"
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-baseline-batch $EXTRA > $O/joint_$tag.json 2> $O/joint_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/joint_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), "gemm ms", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"])
except Exception as e:
    print("$tag failed", e); print(open("$O/joint_$tag.err").read()[-1500:])
PY
}
run new X=1
run new_again X=1
EXTRA="--replicas 1"
run new_one_stream X=1
