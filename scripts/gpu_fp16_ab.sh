# fp16 two-plane GEMM: row-maxima chain on / off, workgroups-per-CU builds (scripts/build_fp16_variants.sh)
O=gpurun_out/r02_fp16ab
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "fp16 or chain or conv2d" > $O/tests.log 2>&1
echo "tests exit $?"; tail -3 $O/tests.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline > $O/joint_$tag.json 2> $O/joint_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/joint_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"], d.get("single_stream_ms_per_step"))
except Exception as e:
    print("$tag failed", e); print(open("$O/joint_$tag.err").read()[-1500:])
PY
}
run wg5_chain X=1
run wg5_scan APS_GEMM_ROWMAX_CHAIN=0
run wg2_chain APS_AMD_LIB=aps_amd/csrc/libaps_amd_wg2.so
run wg5_chain_again X=1
run bd APS_GEMM_SPLIT_LAYOUT=1
