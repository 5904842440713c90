# one default joint bench line (no CPU legs), value / ms / frac / one-stream ms; extra args -> bench.py
mkdir -p gpurun_out/r02_split
tag=${TAG:-run}
timeout 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r02_split/once_$tag.json 2> gpurun_out/r02_split/once_$tag.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02_split/once_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("single_stream_ms_per_step"))
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/r02_split/once_$tag.err").read()[-800:])
PY
