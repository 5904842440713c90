#!/bin/bash
# Round 5, visit 3: what does the chained launch do?  Tiny cases first, every step under its own short timeout.
set -u
O=gpurun_out/r05_s3
mkdir -p $O
for M in 32 64 2016; do
  timeout 60 python scripts/chain_smoke.py $M > $O/smoke_$M.txt 2>&1; echo "M=$M rc=$?" >> $O/smoke_$M.txt
  cat $O/smoke_$M.txt
done
timeout 60 python scripts/chain_smoke.py 2016 8 > $O/smoke_2016_wg8.txt 2>&1; echo "rc=$?" >> $O/smoke_2016_wg8.txt
cat $O/smoke_2016_wg8.txt
