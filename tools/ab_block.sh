#!/bin/bash
# Placement A/B of the one-wave component-parallel kernels (gpurun -- bash tools/ab_block.sh [tag]): workgroups of 64 / 128 / 256
# threads (MF_CP_BLOCK) against the dispatcher's own rule (mf_common.h::wave_unit_block), forward (recording) and backward
R=gpurun_out/${1:-r4}_ab_block.txt; : > $R
run() { name=$1; shift; echo "# $name" >> $R; env AB_ONLY_CP=1 AB_BWD=1 "$@" timeout 300 python tools/ab_cp.py 2> /dev/null | grep "^B " >> $R; }
run "default dispatch, the library's rule" AB_B=1024,2048,3072,4096,6144,8192,16384
for blk in 64 128 256; do
run "record read by the computing wave (MF_CP_BWD_MODE=2), workgroups of $blk threads" MF_CP_BWD_MODE=2 MF_CP_BLOCK=$blk AB_B=1024,2048,3072,4096,6144,8192
run "recompute, no record (MF_CP_RECORD_MAX_WAVES=0), workgroups of $blk threads" MF_CP_RECORD_MAX_WAVES=0 MF_CP_BLOCK=$blk AB_B=2048,3072,4096,8192
done
cat $R
