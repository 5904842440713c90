# kernel table of the joint TRAINING step (bench.py --workload train) -> gpurun_out/r03_train_prof
O=$GRAFT_REPO_ROOT/gpurun_out/r03_train_prof
mkdir -p $O
export TMPDIR=/tmp
cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $GRAFT_REPO_ROOT/bench.py --workload train --steps 5 --warmup 2 > $O/bench.json 2>&1
cd $GRAFT_REPO_ROOT
rm -f $O/*kernel_trace.csv
head -32 $O/t_kernel_stats.csv | cut -c1-200
tail -c 500 $O/bench.json
