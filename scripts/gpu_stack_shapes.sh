# the layer-pipelined LSTM stack of the 32-utterance batch under other decompositions (APS_LSTM_SHAPE=MT,UT)
for sh in default 2,1 2,2 1,2 1,1; do
  if [ $sh = default ]; then unset APS_LSTM_SHAPE; else export APS_LSTM_SHAPE=$sh; fi
  timeout 250 python bench.py --no-cpu-baseline --group 1 --steps 60 2>/tmp/err.txt | python -c "
import sys, json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('shape $sh:', d['value'], d['ms_per_step'], 'single', d.get('single_stream_ms_per_step'), 'mask_net us', d['stage_us']['mask_net'])
except Exception as e:
    print('shape $sh failed', e); print(open('/tmp/err.txt').read()[-300:])
"
done
