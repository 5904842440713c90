#!/bin/bash
# round 4, visit 18: the LSTM stack's decomposition under the round-4 GEMM (two batches in flight: fewer,
# fatter LSTM workgroups leave more CUs to the other batch's GEMMs)
set -u
O=gpurun_out/r04_s18
mkdir -p $O
for sh in default 2,2 1,2 default; do
  if [ $sh = default ]; then unset APS_LSTM_SHAPE; else export APS_LSTM_SHAPE=$sh; fi
  timeout 250 python bench.py --no-cpu-baseline --group 1 --merged-group 0 --steps 60 --warmup 5 2>$O/err_$sh.txt | python -c "
import sys, json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('shape $sh:', d['value'], d['ms_per_step'], 'single', d.get('single_stream_ms_per_step'), 'mask_net us', d['stage_us'].get('mask_net'))
except Exception as e:
    print('shape $sh failed', e); print(open('$O/err_$sh.txt').read()[-300:])
" | tee -a $O/shapes.txt
done
