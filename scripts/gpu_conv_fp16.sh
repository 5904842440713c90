# the DCCRN blocks on the fp16 two-plane convolution (default for them): model tests, the bench line
O=gpurun_out/r02_conv16
mkdir -p $O
timeout 40 python -m pytest tests/test_gpu_dccrn.py -x -q -m gpu 2>&1 | tail -2
timeout 55 python bench.py --workload dccrn > $O/dccrn_default.json 2> $O/dccrn_default.err
python - <<PY
import json
d = json.loads(open("$O/dccrn_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"][:40], d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"], d.get("parity"))
PY
