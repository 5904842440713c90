mkdir -p gpurun_out/r02_split
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r02_split/j2_$tag.json 2> gpurun_out/r02_split/j2_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02_split/j2_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("single_stream_ms_per_step"))
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/r02_split/j2_$tag.err").read()[-800:])
PY
}
run r2
run r3 --replicas 3
run r4 --replicas 4
run r1 --replicas 1
run g8 --group 8 --batches 6
run g2 --group 2
