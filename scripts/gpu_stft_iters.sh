# STFT: tiles per wavefront (APS_STFT_ITERS) at 32 / 64 / 128 utterances per launch -- stage_roofline.stft of
# the joint bench line
O=gpurun_out/r03_stft_iters
mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --no-cpu-baseline --no-baseline-batch --steps 40 $EXTRA > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    s=d["stage_roofline"]
    print("$tag", d["value"], d["ms_per_step"], "stft us", s["stft"]["us_per_launch"], s["stft"]["frac"], "stage", s["all_stages"]["us_per_batch"], s["all_stages"]["frac"])
except Exception as e:
    print("$tag failed", e); print(open("$O/$tag.err").read()[-800:])
PY
}
EXTRA=""
run g4_iters1 APS_STFT_ITERS=1
run g4_iters2 APS_STFT_ITERS=2
EXTRA="--group 2"
run g2_default X=1
run g2_iters1 APS_STFT_ITERS=1
run g2_iters3 APS_STFT_ITERS=3
run g2_iters4 APS_STFT_ITERS=4
