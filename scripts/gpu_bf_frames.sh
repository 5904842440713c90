# beamform: frames per wavefront (APS_BF_FRAMES) -- stage_roofline.beamform of the joint bench line
O=gpurun_out/r03_bf_frames
mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --no-cpu-baseline --no-baseline-batch --steps 30 $EXTRA > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    s=d["stage_roofline"]
    print("$tag", d["value"], "stft", s["stft"]["us_per_launch"], "feat", s["features"]["us_per_launch"], "mvdr", s["mvdr_weights"]["us_per_launch"], "bf", s["beamform"]["us_per_launch"], s["beamform"]["frac"], "stage", s["all_stages"]["us_per_batch"], s["all_stages"]["frac"])
except Exception as e:
    print("$tag failed", e); print(open("$O/$tag.err").read()[-800:])
PY
}
EXTRA=""
run g4_default X=1
run g4_bf3 APS_BF_FRAMES=3
run g4_bf5 APS_BF_FRAMES=5
EXTRA="--group 1"
run g1_default X=1
run g1_bf3 APS_BF_FRAMES=3
