#!/bin/bash
# rocprofv3 view of the lift-splat kernels at config-4 shapes (tools/bench_lift_splat.py): kernel trace, then one PMC pass per group.
# gpurun -- bash tools/pmc_splat.sh <tag>
TAG=${1:-rX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python tools/bench_lift_splat.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ls -o ls -- $CMD > $OUT/${TAG}_bench_lift_splat_under_rocprof.txt 2> $OUT/prof_ls.err
f=$(find $OUT/prof_ls -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -i -E "Name|splat|lift|bev" "$f" | head -30 > $OUT/${TAG}_lift_splat_kernel_stats.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  n=$(echo $grp | tr ' ' '_')
  rm -rf $OUT/pmc_$n
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$n -o p -- $CMD > /dev/null 2> $OUT/pmc_$n.err
  f=$(find $OUT/pmc_$n -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" | grep -i -E "splat|lift|bev" >> $OUT/${TAG}_pmc_lift_splat.txt; else echo "group [$grp] failed" >> $OUT/${TAG}_pmc_lift_splat.txt; fi
done
cat $OUT/${TAG}_lift_splat_kernel_stats.csv $OUT/${TAG}_pmc_lift_splat.txt
