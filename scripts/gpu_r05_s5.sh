#!/bin/bash
set -u
O=gpurun_out/r05_s5
mkdir -p $O
timeout 45 python scripts/chain_debug.py 2016 3 > $O/dbg_2016_3.txt 2>&1; echo "rc=$?" >> $O/dbg_2016_3.txt
grep -v amdgpu.ids $O/dbg_2016_3.txt
if ! grep -q "finished=True" $O/dbg_2016_3.txt; then echo "chain still stuck: stopping"; exit 1; fi
for a in "2016" "2016 8" "2016 1024" "70"; do
  n=$(echo $a | tr ' ' '_')
  timeout 60 python scripts/chain_smoke.py $a > $O/smoke_$n.txt 2>&1; echo "rc=$?" >> $O/smoke_$n.txt
  echo "== smoke $a"; grep -v amdgpu.ids $O/smoke_$n.txt
done
timeout 300 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "chain" 2>&1 | tail -8 > $O/pytest_chain.txt
cat $O/pytest_chain.txt
timeout 300 python -m pytest tests/test_gpu_joint.py tests/test_gpu_cplx.py -x -q -m gpu 2>&1 | tail -8 > $O/pytest_joint_cplx.txt
cat $O/pytest_joint_cplx.txt
