#!/bin/bash
# Several PMC passes over the main workload's kernels for one library: bash tools/pmc_sets.sh <tag> [bench.py arguments]   (GTX_LIB selects the build)
# Prints the position-hinted pass' line of every pass.
tag=${1:-x}; shift
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_IFETCH" \
           "SQ_WAVES SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" \
           "SQ_WAVES SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_WR"; do
  i=$((i+1))
  GTX_PMC="$set" bash tools/pmc_main.sh ${tag}_$i "$@" 2>&1 | grep -i "${GTX_PMC_KERNEL:-hinted}"
done
