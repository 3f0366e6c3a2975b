#!/bin/bash
# counters of the rollout kernels that run at saturated batch sizes (c3 step: states-only forward + backward),
# one rocprofv3 --pmc pass per group.   gpurun -- bash tools/pmc_backward_sat.sh "16384 8192" > profile text under gpurun_out/
for Bn in ${1:-16384 8192}; do
  echo "=== c3 step at B = $Bn (T = 500, N = 4, shared 256x256 map pair) ==="
  bash tools/pmc_groups.sh $Bn c3 \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
    "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES" \
    "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" \
    "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS" \
    "TA_BUSY TCP_PENDING_STALL_CYCLES TA_FLAT_WAVEFRONTS" \
    "TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TOTAL_ACCESSES TCP_TOTAL_CACHE_ACCESSES" \
    "TCP_TCC_ATOMIC_WITH_RET_REQ TCP_TCC_ATOMIC_WITHOUT_RET_REQ" \
    "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_FLAT" \
    "FETCH_SIZE" "WRITE_SIZE"
done
