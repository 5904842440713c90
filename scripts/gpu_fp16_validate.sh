# the fp16 two-plane GEMM as the DEFAULT: whole parity suite, smoke, replay-vs-eager soak with two
# and three graphs in flight, one-stream kernel table of the joint step
set -u
O=gpurun_out/r02_fp16v
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x --tb=short > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
REPLICA_DIFF_ROUNDS=100 timeout 400 python scripts/replica_diff.py 2 4 > $O/soak_2x4.log 2>&1; echo "soak 2x4 exit $?"; grep -c differ $O/soak_2x4.log; tail -1 $O/soak_2x4.log | cut -c1-200
REPLICA_DIFF_ROUNDS=60 timeout 400 python scripts/replica_diff.py 2 1 > $O/soak_2x1.log 2>&1; echo "soak 2x1 exit $?"; grep -c differ $O/soak_2x1.log; tail -1 $O/soak_2x1.log | cut -c1-200
REPLICA_DIFF_ROUNDS=40 timeout 400 python scripts/replica_diff.py 3 3 > $O/soak_3x3.log 2>&1; echo "soak 3x3 exit $?"; grep -c differ $O/soak_3x3.log; tail -1 $O/soak_3x3.log | cut -c1-200
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_joint1 -o trace -- \
   python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --replicas 1 > $R/$O/bench_joint1_under_rocprof.json 2>&1)
python scripts/trace_by_grid.py $(find $O/prof_joint1 -name "*kernel_trace.csv" | head -1) > $O/by_grid.txt 2>/dev/null; head -24 $O/by_grid.txt | cut -c1-170
rm -f $O/prof_joint1/*kernel_trace.csv
