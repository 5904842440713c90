#!/bin/bash
# round 4, visit 24: the LSTM stack at 240 VGPRs (one gather register set for both interleaved groups) against
# the 299-register build (libaps_amd_lstm299.so): parity, then the bench A/B on one box
set -u
O=gpurun_out/r04_s24; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_joint.py tests/test_gpu_replicas.py -x -q -m gpu -k "lstm or rnn or joint or replica" 2>&1 | tail -5 > $O/pytest.txt; tail -3 $O/pytest.txt
for lib in lstm299 shipped lstm299 shipped; do
  if [ $lib = shipped ]; then unset APS_AMD_LIB; else export APS_AMD_LIB=$PWD/aps_amd/csrc/libaps_amd_$lib.so; fi
  timeout 300 python bench.py --no-cpu-baseline --group 1 --merged-group 0 --steps 60 --warmup 5 2>$O/err_$lib.txt | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib:', d['value'], d['ms_per_step'], 'single', d.get('single_stream_ms_per_step'), 'mask_net us', d['stage_us'].get('mask_net'), 'timeouts', d.get('lstm_handoff_timeouts'))
" | tee -a $O/ab.txt
done
unset APS_AMD_LIB
timeout 300 python scripts/gemm_sequence_overlap.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/gemm_sequence_overlap.txt
