for r in 2 1; do for g in 4 8 16; do timeout 250 python bench.py --no-cpu-baseline --no-baseline-batch --replicas $r --group $g 2>/tmp/err.txt | python -c "
import sys, json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('replicas $r group $g:', d['value'], d['ms_per_step'], d.get('single_stream_ms_per_step'), 'gemm frac', d['roofline']['frac'], 'gemm ms', d['roofline'].get('kernel_ms_per_step'))
except Exception as e:
    print('replicas $r group $g failed', e); print(open('/tmp/err.txt').read()[-600:])
"; done; done
