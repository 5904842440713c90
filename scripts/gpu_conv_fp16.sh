# the convolutions on the fp16 two-plane arithmetic: parity tests, DCCRN with and without
O=gpurun_out/r02_conv16
mkdir -p $O
timeout 60 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "conv2d_nhwc and fp16x2" 2>&1 | tail -3
timeout 40 python bench.py --workload dccrn --no-cpu-baseline > $O/dccrn_bf16.json 2> $O/dccrn_bf16.err
APS_CONV_FP16X2=1 timeout 60 python bench.py --workload dccrn > $O/dccrn_fp16.json 2> $O/dccrn_fp16.err
python - <<PY
import json
for tag in ("bf16", "fp16"):
    try:
        d = json.loads(open("$O/dccrn_%s.json" % tag).read().strip().splitlines()[-1])
        print(tag, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"], d.get("parity"))
    except Exception as e:
        print(tag, "failed", e); print(open("$O/dccrn_%s.err" % tag).read()[-800:])
PY
