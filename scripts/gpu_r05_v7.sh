#!/bin/bash
set -u
O=gpurun_out/r05_v7
mkdir -p $O
for w in own null head worker; do
APS_HOST_INPUT_STREAM=$w timeout 300 python bench.py --no-cpu-baseline --merged-group 0 2> $O/bench_$w.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d.get('host_input'); print('$w', d['value'], d['ms_per_step'], h.get('value'), h.get('ms_per_step'), h.get('error'))"
done
python - <<'PY'
import torch, time
# raw H2D rate of one 32.8 MB pinned buffer on an otherwise idle GPU
x = torch.empty(32, 4, 64000).pin_memory(); y = torch.empty_like(x, device="cuda")
torch.cuda.synchronize()
for _ in range(3): y.copy_(x, non_blocking=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): y.copy_(x, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print("H2D 32.8 MB pinned:", round(dt * 1e3, 3), "ms =", round(x.numel() * 4 / dt / 1e9, 1), "GB/s")
PY
