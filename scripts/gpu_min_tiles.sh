# where the fp16 two-plane GEMM starts to pay at BASELINE's 32-per-GPU batch (M = 2016): bench --group 1
# under different SPLIT_MIN_TILES (APS_SPLIT_MIN_TILES) and K bounds
for t in 320 250 190 120 60; do
  APS_GEMM_SPLIT_MIN_TILES=$t timeout 250 python bench.py --no-cpu-baseline --group 1 --steps 60 2>/tmp/err.txt | python -c "
import sys, json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    print('min tiles $t:', d['value'], d['ms_per_step'], 'single', d.get('single_stream_ms_per_step'), r['kernel'][:40], r['frac'], r.get('kernel_ms_per_step'), 'other', r.get('other_gemm_kernels'))
except Exception as e:
    print('min tiles $t failed', e); print(open('/tmp/err.txt').read()[-500:])
"
done
