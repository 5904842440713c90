#!/bin/bash
# PMC passes over one GEMM variant: gpu_split_pmc.sh <tag> <split 0|1> <M> <N> <K> [kernel substring]
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
sub=${5:-gemm_}
out=$R/gpurun_out/r02_split/pmc_$tag
mkdir -p $out
cd /tmp
run() {
  name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -o p -- \
     python $R/scripts/split_pmc_target.py $ARGS > $out/$name.log 2>&1
}
ARGS="$1 $2 $3 $4"
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA
run c SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM
run d TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum
python $R/scripts/pmc_kernel_table.py "$sub" $(find $out -name "*counter_collection.csv") > $out/table.txt 2>&1
cat $out/table.txt
