#!/bin/bash
# round 4, visit 22: the LSTM kernels' slow path re-requests only the missing chunks: parity, the GEMM
# sequence beside a running LSTM, the bench lines
set -u
O=gpurun_out/r04_s22; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_joint.py tests/test_gpu_dccrn.py -x -q -m gpu -k "lstm or rnn or joint or dccrn or mask" 2>&1 | tail -5 > $O/pytest.txt; tail -3 $O/pytest.txt
timeout 300 python scripts/gemm_sequence_overlap.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_sequence_overlap.txt | tail -4
for i in 1 2; do
timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2> $O/bench_joint.err | tail -1 > $O/bench_joint_$i.json
python - <<PY
import json
d=json.load(open("gpurun_out/r04_s22/bench_joint_$i.json"))
print("joint: value", d["value"], "ms", d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), "mask_net us", d["stage_us"].get("mask_net"), "merged", d["merged_batch"]["value"], "timeouts", d.get("lstm_handoff_timeouts"))
PY
done
