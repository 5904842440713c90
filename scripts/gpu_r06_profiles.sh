#!/bin/bash
# Round 6: the rocprof evidence under the bench lines, from the shipped tree, ONE configuration and ONE stream per trace:
#   kernel traces   joint at 32 utterances per launch as the LIBRARY DEFAULT runs it on one stream (per-launch
#                   projections), the same step with the conformer stack as one launch per batch (APS_MEGA=1: the
#                   kernel the headline's six worker streams run; its average duration is what `roofline.per_launch.ms`
#                   of the bench line must agree with), the front end
#   PMC passes      MFMA busy and FETCH_SIZE / WRITE_SIZE (separate passes) for the one-launch-per-batch step
set -u
R=$(pwd)
O=gpurun_out/r06_prof
mkdir -p $O
export TMPDIR=/tmp
trace() {  # name, env assignments (quoted, may be empty), bench arguments...
  local n=$1 e=$2; shift 2
  (cd /tmp && env $e timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr_$n -o t -- \
     python $R/bench.py "$@" --no-cpu-baseline > $R/$O/tr_$n.log 2>&1)
  local f=$(find $O/tr_$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/${n}_kernel_stats.csv
  grep '^{"metric"' $O/tr_$n.log | tail -1 > $O/${n}_line_under_rocprof.json
  rm -rf $O/tr_$n
}
trace joint32_one_stream "APS_MEGA=0" --group 1 --merged-group 0 --replicas 1 --pipeline 0 --steps 40 --warmup 5
trace joint32_one_stream_conformer_stack "APS_MEGA=1" --group 1 --merged-group 0 --replicas 1 --pipeline 0 --steps 40 --warmup 5
trace frontend_one_stream "APS_X=1" --workload frontend --replicas 1 --steps 60 --warmup 5
# ... and the DEFAULT command itself (the pipelined headline: the conformer-stack launches of six batches beside the LSTM
# launches and the front ends -- the kernel's average duration here is its duration UNDER LOAD, `stage_ms_under_load`)
trace joint_default_pipeline "APS_X=1" --no-host-input
pmc() {  # name, counters (quoted), env, bench arguments...
  local n=$1 c=$2 e=$3; shift 3
  (cd /tmp && env $e timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc_$n -o p -- \
     python $R/bench.py "$@" --eager --repeats 1 --no-cpu-baseline > $R/$O/pmc_$n.log 2>&1)
}
M="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"
pmc mfma32 "$M" "APS_MEGA=1" --group 1 --merged-group 0 --replicas 1 --pipeline 0 --steps 3 --warmup 2
f=$(find $O/pmc_mfma32 -name "p_counter_collection.csv" | head -1)
[ -n "$f" ] && python scripts/pmc_mfma_summary.py $f > $O/joint32_conformer_stack_pmc_mfma.csv
for c in FETCH_SIZE WRITE_SIZE; do
  pmc j32_$c $c "APS_MEGA=1" --group 1 --merged-group 0 --replicas 1 --pipeline 0 --steps 5 --warmup 2
done
f1=$(find $O/pmc_j32_FETCH_SIZE -name "p_counter_collection.csv" | head -1)
f2=$(find $O/pmc_j32_WRITE_SIZE -name "p_counter_collection.csv" | head -1)
[ -n "$f1" ] && [ -n "$f2" ] && python scripts/pmc_traffic_summary.py $f1 $f2 > $O/joint32_conformer_stack_pmc_traffic_raw.csv
W="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
pmc wait32 "$W" "APS_MEGA=1" --group 1 --merged-group 0 --replicas 1 --pipeline 0 --steps 3 --warmup 2
f=$(find $O/pmc_wait32 -name "p_counter_collection.csv" | head -1)
[ -n "$f" ] && python scripts/pmc_summary.py $f > $O/joint32_conformer_stack_pmc_wait.csv
# the front end's kernels: what the fused STFT + feature kernel waits for (two passes of 8 counters)
FA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
FB="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES SQ_WAVES"
pmc fe_A "$FA" "APS_X=1" --workload frontend --replicas 1 --steps 6 --warmup 2
pmc fe_B "$FB" "APS_X=1" --workload frontend --replicas 1 --steps 6 --warmup 2
files=$(find $O/pmc_fe_A $O/pmc_fe_B -name "p_counter_collection.csv" | sort)
[ -n "$files" ] && python scripts/pmc_summary.py $files > $O/frontend_pmc_wait_valu.csv
find $O -name "*.db" -delete; rm -rf $O/pmc_*/ ; 
head -8 $O/joint32_one_stream_conformer_stack_kernel_stats.csv | cut -c1-150
head -6 $O/joint32_conformer_stack_pmc_mfma.csv | cut -c1-160
head -8 $O/joint32_conformer_stack_pmc_traffic_raw.csv | cut -c1-160
head -4 $O/joint32_conformer_stack_pmc_wait.csv | cut -c1-300
head -3 $O/frontend_pmc_wait_valu.csv | cut -c1-400
