#!/bin/bash
set -u
O=gpurun_out/r05_v3
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_joint.py tests/test_gpu_train.py -q -m gpu -x -k "mvdr or joint or enh or config" 2>&1 | tail -6
timeout 200 python scripts/step_torch_ops.py 2>&1 | tail -9
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --merged-group 0 2> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'one', d.get('single_stream_ms_per_step'), 'stages', d['stage_roofline']['all_stages']['survey_8d']['frac'], {k: (v.get('us_per_launch'), v.get('frac')) for k, v in d['stage_roofline'].items() if isinstance(v, dict) and 'frac' in v})"
done
