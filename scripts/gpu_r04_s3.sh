#!/bin/bash
# round 4, visit 3: where the 32-utterance step goes -- kernel trace of the replayed step, one stream
set -u
O=gpurun_out/r04_s3
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lay in 3 2; do
(cd /tmp && APS_GEMM_SPLIT_LAYOUT=$lay timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_g1_l$lay -o trace -- \
   python $R/bench.py --group 1 --replicas 1 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline > $R/$O/bench_g1_l$lay.json 2> $R/$O/bench_g1_l$lay.err)
f=$(find $O/prof_g1_l$lay -name "*kernel_stats.csv" | head -1)
cp $f $O/g1_l${lay}_kernel_stats.csv
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print(sys.argv[1], "total ms", tot/1e6)
for r in rows[:22]:
    print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):6d} total {float(r["TotalDurationNs"])/1e6:9.2f} ms avg {float(r["AverageNs"])/1e3:8.1f} us {float(r["Percentage"]):5.1f}%')
PY
tail -c 600 $O/bench_g1_l$lay.json | head -c 300; echo
done
