#!/bin/bash
# What each stage of the record-reading backward (MODE = kCpSaved) costs at 2048 ... 8192 rollouts: A/B builds with one stage
# compiled out (results are wrong; only the time is read).  Build here, run on the GPU box:
#   tools/ab_saved_variants.sh build ;  gpurun -- bash tools/ab_saved_variants.sh run
cd "$(dirname "$0")/.."
VARS="GATHER STATE REC UP ATOMIC"
if [ "$1" = build ]; then
  for v in $VARS; do f="-DMF_SAVED_NO_$v"; [ $v = GATHER ] && f="-DMF_STREAM_NO_GATHER"      # (the cells as constants: left undefined they are NaN work)
    tools/build_variant.sh saved_no_$v "$f" rollout_bwd_cp_fast.hip; done
  tools/build_variant.sh saved_no_mem "-DMF_STREAM_NO_GATHER -DMF_SAVED_NO_STATE -DMF_SAVED_NO_REC -DMF_SAVED_NO_UP -DMF_SAVED_NO_ATOMIC" rollout_bwd_cp_fast.hip
  tools/build_variant.sh saved_no_alu "-DMF_SAVED_NO_VJP -DMF_SAVED_NO_REBUILD" rollout_bwd_cp_fast.hip
  exit 0
fi
R=gpurun_out/${2:-r4}_ab_saved_variants.txt; : > $R
export AB_ONLY_CP=1 AB_BWD=1 AB_B=${AB_B:-3072,4096} MF_CP_BWD_MODE=2
echo "# product" >> $R; timeout 300 python tools/ab_cp.py 2> /dev/null | grep "^B " >> $R
for v in $VARS mem alu; do
  lv=$v
  echo "# without $v" >> $R
  MONOFORCE_HIP_LIB=$PWD/gpurun_in_ab/saved_no_$lv/libmonoforce_hip.so timeout 300 python tools/ab_cp.py 2> /dev/null | grep "^B " >> $R
done
cat $R
