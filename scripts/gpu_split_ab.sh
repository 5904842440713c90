mkdir -p gpurun_out/r02_split
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02_split/joint_$tag.json 2> gpurun_out/r02_split/joint_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02_split/joint_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("stats"))
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/r02_split/joint_$tag.err").read()[-1500:])
PY
}
run fp32 APS_GEMM_SPLIT=0
run auto X=1
run v1 APS_SPLIT_KERNEL=v1
run pc APS_SPLIT_KERNEL=pc
run v1_128 APS_SPLIT_KERNEL=v1 APS_SPLIT_TN=128
