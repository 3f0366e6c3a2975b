#!/bin/bash
# The mid-range backward (B = 2048 .. 8192, N = 4, default integrator) under every backward form and gradient-copy count.
# gpurun -- bash tools/ab_midrange.sh <tag>   -> gpurun_out/<tag>/<tag>_ab_midrange.txt
TAG=${1:-rX}; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$OUT/${TAG}_ab_midrange.txt; : > $R
run() { name=$1; shift; echo "# $name" >> $R; env AB_ONLY_CP=1 AB_BWD=1 AB_B=${AB_B:-2048,4096,8192} "$@" timeout 300 python tools/ab_cp.py 2> /dev/null | grep "^B " >> $R; }
run "default dispatch (stream <= 512 workgroups, record read by the computing wave <= 1024, recompute early beyond)"
run "record read by the computing wave (MF_CP_BWD_MODE=2)" MF_CP_BWD_MODE=2
run "recompute, late (no record)" MF_CP_BWD_MODE=1 MF_CP_RECORD_MAX_WAVES=0
run "recompute, early (no record)" MF_CP_BWD_MODE=0 MF_CP_RECORD_MAX_WAVES=0
run "default dispatch, 128 gradient copies" MF_GRAD_COPIES=128
run "default dispatch, 16 gradient copies" MF_GRAD_COPIES=16
run "streaming up to 1024 workgroups (6-slot ring: the third and fourth workgroup of a CU wait for LDS)" MF_CP_STREAM_MAX_GRID=1024
cat $R
