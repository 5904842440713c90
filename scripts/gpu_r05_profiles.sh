#!/bin/bash
# Round 5: the rocprof evidence under the bench lines, from the shipped tree, ONE configuration and ONE stream per
# trace (kernel averages are then not mixed across batch sizes or stretched by a second stream):
#   kernel traces   joint at 32 utterances per launch as the LIBRARY DEFAULT runs it (one stream: the one-tile-per-CU
#                   projections on the K-group form), the same with APS_GEMM_KGROUP=0 (= what each of the headline's
#                   two streams runs: four-wave panel tiles throughout), the front end
#   PMC passes      MFMA busy (joint 32, default), FETCH_SIZE and WRITE_SIZE in separate passes (joint 32, default)
set -u
R=$(pwd)
O=gpurun_out/r05_prof
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
trace joint32_one_stream "APS_X=1" --group 1 --merged-group 0 --replicas 1 --steps 40 --warmup 5
trace joint32_one_stream_panel_only "APS_GEMM_KGROUP=0" --group 1 --merged-group 0 --replicas 1 --steps 40 --warmup 5
trace frontend_one_stream "APS_X=1" --workload frontend --replicas 1 --steps 60 --warmup 5
pmc() {  # name, counters (quoted), bench arguments...
  local n=$1 c=$2; shift 2
  (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc_$n -o p -- \
     python $R/bench.py "$@" --eager --repeats 1 --no-cpu-baseline > $R/$O/pmc_$n.log 2>&1)
}
M="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"
pmc mfma32 "$M" --group 1 --merged-group 0 --replicas 1 --steps 3 --warmup 2
python scripts/pmc_mfma_summary.py $O/pmc_mfma32/p_counter_collection.csv > $O/joint32_pmc_mfma.csv
for c in FETCH_SIZE WRITE_SIZE; do
  pmc j32_$c $c --group 1 --merged-group 0 --replicas 1 --steps 5 --warmup 2
done
python scripts/pmc_traffic_summary.py $O/pmc_j32_FETCH_SIZE/p_counter_collection.csv $O/pmc_j32_WRITE_SIZE/p_counter_collection.csv > $O/joint32_pmc_traffic_raw.csv
rm -rf $O/pmc_*/p_kernel_trace.csv $O/pmc_*/p_counter_collection.csv $O/pmc_*/*.db
head -14 $O/joint32_one_stream_kernel_stats.csv | cut -c1-150
head -8 $O/joint32_one_stream_panel_only_kernel_stats.csv | cut -c1-150
cat $O/joint32_pmc_mfma.csv | head -8 | cut -c1-160
cat $O/joint32_pmc_traffic_raw.csv | head -12 | cut -c1-160
