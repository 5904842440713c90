#!/bin/bash
# round 4, visit 23: the LSTM stack bounded to 256 VGPRs (two waves per SIMD: spills, but a second batch's
# GEMM keeps two workgroups per CU beside it instead of one) against the shipped 299-register build
set -u
O=gpurun_out/r04_s23; mkdir -p $O
for lib in shipped lstmocc2 shipped lstmocc2; do
  if [ $lib = shipped ]; then unset APS_AMD_LIB; else export APS_AMD_LIB=$PWD/aps_amd/csrc/libaps_amd_$lib.so; fi
  timeout 300 python bench.py --no-cpu-baseline --group 1 --merged-group 0 --steps 60 --warmup 5 2>$O/err_$lib.txt | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib:', d['value'], d['ms_per_step'], 'single', d.get('single_stream_ms_per_step'), 'mask_net us', d['stage_us'].get('mask_net'), 'timeouts', d.get('lstm_handoff_timeouts'))
" | tee -a $O/ab.txt
done
