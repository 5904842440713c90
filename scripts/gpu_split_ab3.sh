mkdir -p gpurun_out/r02_split
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02_split/j4_$tag.json 2> gpurun_out/r02_split/j4_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02_split/j4_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("single_stream_ms_per_step"))
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/r02_split/j4_$tag.err").read()[-800:])
PY
}
run bd APS_GEMM_SPLIT_LAYOUT=1
run v1 APS_GEMM_SPLIT_LAYOUT=0
run bd2 APS_GEMM_SPLIT_LAYOUT=1
