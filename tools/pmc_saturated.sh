#!/bin/bash
# SQ / TCC / TA counters of the forward kernel at a given batch (one rocprofv3 --pmc pass per counter group).
#   gpurun -- bash tools/pmc_saturated.sh 65536 [workload]
Bn=${1:-65536}
W=${2:-c3f}
OUT=gpurun_out/pmc_sat_$Bn
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" "TCC_EA_WRREQ_STALL TCC_EA_WRREQ TCC_EA_WRREQ_64B TCC_BUSY" \
           "TA_BUSY TCP_PENDING_STALL_CYCLES TA_BUFFER_WAVEFRONTS TA_FLAT_WAVEFRONTS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o p -- python bench.py --steps 3 --warmup 1 --workload $W --batch $Bn --no-cpu-baseline --no-others > /dev/null 2> $OUT/g$i.err
  f=$(find $OUT/g$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" rollout; else echo "group $i failed: $(tail -2 $OUT/g$i.err)"; fi
done
