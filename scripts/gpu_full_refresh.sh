#!/bin/bash
# Full GPU visit: parity suite, smoke, every bench workload, kernel-trace profiles of the joint and
# DCCRN workloads.  Outputs -> gpurun_out/ (copy what is to be judged into profiles/).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu =="
timeout 1200 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
echo "== smoke =="
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee gpurun_out/smoke.log
echo "== bench (default: joint) =="
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_joint.json
for w in frontend encoder dccrn; do
  echo "== bench $w =="
  timeout 600 python bench.py --workload $w 2>&1 | tail -1 | tee gpurun_out/bench_$w.json
done
echo "== rocprofv3 joint (one graph on one stream: kernel durations without a second batch beside them) =="
rm -rf gpurun_out/prof_joint gpurun_out/prof_dccrn
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_joint -o trace -- \
   python $R/bench.py --steps 50 --warmup 12 --no-cpu-baseline --replicas 1 > $R/gpurun_out/bench_joint_under_rocprof.json 2>&1)
head -12 $(find gpurun_out/prof_joint -name "*kernel_stats.csv" | head -1) | cut -c1-160
echo "== rocprofv3 dccrn =="
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dccrn -o trace -- \
   python $R/bench.py --workload dccrn --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_dccrn_under_rocprof.json 2>&1)
head -8 $(find gpurun_out/prof_dccrn -name "*kernel_stats.csv" | head -1) | cut -c1-160
echo "== rocprofv3 --pmc (MFMA busy) joint, eager =="
rm -rf gpurun_out/pmc_joint
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv \
   -d $R/gpurun_out/pmc_joint -o p -- python $R/bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_joint.log 2>&1)
python scripts/pmc_mfma_summary.py gpurun_out/pmc_joint/p_counter_collection.csv | tee gpurun_out/joint_pmc_mfma.csv | head -8
echo "== rocprofv3 encoder =="
rm -rf gpurun_out/prof_encoder
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_encoder -o trace -- \
   python $R/bench.py --workload encoder --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_encoder_under_rocprof.json 2>&1)
head -8 $(find gpurun_out/prof_encoder -name "*kernel_stats.csv" | head -1) | cut -c1-160
