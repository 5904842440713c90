#!/bin/bash
# Full GPU visit (round 3): parity suite, smoke, every bench workload, kernel-trace profiles, PMC passes.
# Outputs -> gpurun_out/r03_full/ (copy what is to be judged into profiles/).
set -u
O=gpurun_out/r03_full
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu =="
timeout 1200 python -m pytest tests -m gpu -q -s --tb=short > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
echo "== smoke =="
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee $O/smoke.log
echo "== bench (default: joint) =="
timeout 600 python bench.py --steps 20 --warmup 5 2> $O/bench_joint.err | tail -1 > $O/bench_joint.json; cut -c1-300 $O/bench_joint.json
for w in frontend encoder dccrn train; do
  echo "== bench $w =="
  timeout 600 python bench.py --workload $w 2> $O/bench_$w.err | tail -1 > $O/bench_$w.json; cut -c1-200 $O/bench_$w.json
done
echo "== bench joint --replicas 1 / --group 1 =="
timeout 300 python bench.py --replicas 1 --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_joint_r1.json
timeout 300 python bench.py --group 1 --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_joint_g1.json
echo "== rocprofv3 joint, default command (two batches in flight) =="
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_joint -o trace -- \
   python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/$O/bench_joint_under_rocprof.json 2>&1)
head -8 $(find $O/prof_joint -name "*kernel_stats.csv" | head -1) | cut -c1-140
echo "== rocprofv3 joint, one stream (kernel durations without a second batch beside them) =="
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_joint1 -o trace -- \
   python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --replicas 1 > $R/$O/bench_joint1_under_rocprof.json 2>&1)
echo "== rocprofv3 frontend =="
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_frontend -o trace -- \
   python $R/bench.py --workload frontend --steps 100 --no-cpu-baseline > $R/$O/bench_frontend_under_rocprof.json 2>&1)
head -8 $(find $O/prof_frontend -name "*kernel_stats.csv" | head -1) | cut -c1-140
echo "== rocprofv3 --pmc (MFMA busy) joint, eager =="
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv \
   -d $R/$O/pmc_joint -o p -- python $R/bench.py --eager --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline > $R/$O/pmc_joint.log 2>&1)
python scripts/pmc_mfma_summary.py $O/pmc_joint/p_counter_collection.csv > $O/joint_pmc_mfma.csv; head -8 $O/joint_pmc_mfma.csv | cut -c1-160
echo "== rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, frontend (separate passes) =="
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc_$c -o p -- \
     python $R/bench.py --workload frontend --eager --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline > $R/$O/pmc_$c.log 2>&1)
done
python scripts/pmc_traffic_summary.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv > $O/pmc_traffic_raw.csv; cat $O/pmc_traffic_raw.csv | cut -c1-160
rm -rf $O/prof_joint/*kernel_trace.csv $O/prof_joint1/*kernel_trace.csv $O/prof_frontend/*kernel_trace.csv $O/pmc_*/p_kernel_trace.csv
ls $O
