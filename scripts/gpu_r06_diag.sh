#!/bin/bash
# Round 6, first visit: (1) the panel GEMM's workgroup life alone / beside two other streams / in the headline pipeline
# (trace build), (2) PMC passes naming what the panel kernel and the fused STFT kernel wait for, (3) a default bench line
# from this round's starting tree.
set -u
R=$(pwd)
O=gpurun_out/r06_diag
mkdir -p $O
export TMPDIR=/tmp
APS_AMD_LIB=$R/aps_amd/csrc/libaps_amd_ptrace.so timeout 600 python scripts/panel_trace_under_load.py > $O/panel_trace_under_load.txt 2>&1
tail -5 $O/panel_trace_under_load.txt
(cd /tmp && rocprofv3 -L > $R/$O/counters_list.txt 2>&1)
grep -c . $O/counters_list.txt
pmc() {  # name, counters (quoted), bench arguments...
  local n=$1 c=$2; shift 2
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc_$n -o p -- \
     python $R/bench.py "$@" --eager --repeats 1 --no-cpu-baseline > $R/$O/pmc_$n.log 2>&1)
  ls $O/pmc_$n/*/p_counter_collection.csv $O/pmc_$n/p_counter_collection.csv 2>/dev/null | head -1
}
J="--group 1 --merged-group 0 --replicas 1 --pipeline 0 --steps 3 --warmup 2"
F="--workload frontend --replicas 1 --steps 6 --warmup 2"
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
B="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE"
C="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
D="TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_DRAM_sum GRBM_GUI_ACTIVE"
E="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
for s in A B C D E; do
  pmc j_$s "${!s}" $J
  pmc f_$s "${!s}" $F
done
for w in j f; do
  files=$(find $O/pmc_${w}_* -name "p_counter_collection.csv" | sort)
  [ -n "$files" ] && python scripts/pmc_summary.py $files > $O/pmc_${w}_summary.csv
done
for d in $O/pmc_*; do [ -d $d ] && tail -3 $d.log | cut -c1-200 > $d.tail; done
find $O -name "*.db" -delete; find $O -name "p_kernel_trace.csv" -delete; find $O -name "p_counter_collection.csv" -size +20M -delete
python bench.py > $O/bench_default.log 2>&1
grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
cut -c1-600 $O/bench_default.json
cat $O/pmc_j_summary.csv | head -6 | cut -c1-400
