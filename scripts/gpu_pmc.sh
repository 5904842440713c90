#!/bin/bash
# PMC passes over the bench step (separate runs per counter group, kernel-trace only).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
run() { # name, counters...
  name=$1; shift
  rm -rf $R/gpurun_out/pmc_$name
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$name -o p -- \
     python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline  > $R/gpurun_out/pmc_$name.log 2>&1
  ls $R/gpurun_out/pmc_$name | head -3
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
