# fp16 two-plane GEMM: tile order experiments (libaps_amd_ord.so built with -DAPS_FP16X2_ORDER_EXPERIMENT
# -DAPS_FP16X2_MIN_WG=4): 0 = row panel major with rotated K start (shipped), 1 = row panels fastest,
# 2 = the workgroups expected on one CU share their weight columns
O=gpurun_out/r02_order
mkdir -p $O
export APS_AMD_LIB=aps_amd/csrc/libaps_amd_ord.so
for o in 0 2; do
  echo "== order $o"
  SPLIT_BENCH_ONLY=fp16 APS_FP16X2_ORDER=$o timeout 200 python scripts/split_gemm_bench.py 8064 2>/dev/null | cut -c1-120
done
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline > $O/joint_$tag.json 2> $O/joint_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/joint_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d.get("single_stream_ms_per_step"))
except Exception as e:
    print("$tag failed", e); print(open("$O/joint_$tag.err").read()[-1500:])
PY
}
run order0 APS_FP16X2_ORDER=0

run order2 APS_FP16X2_ORDER=2
run order0_again APS_FP16X2_ORDER=0
run order2_again APS_FP16X2_ORDER=2
