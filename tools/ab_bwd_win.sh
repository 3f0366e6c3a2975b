#!/bin/bash
# Saturated backward of the 4-point body: cell gradients through an LDS window per workgroup (rollout_bwd_kernel.h WIN)
# against the register accumulators + device-scope atomics, and the same kernel with its atomics compiled out (the bound).
#   tools/build_variant.sh xs_noatomic "-DMF_NO_ATOMICS" rollout_bwd_xs_fast.hip ;  gpurun -- bash tools/ab_bwd_win.sh
cd "$(dirname "$0")/.."
R=gpurun_out/${1:-r5}_ab_bwd_win.txt; : > $R
run() { echo "# $1" >> $R; shift; env "$@" timeout 300 python tools/ab_mw_small.py $BS 2> /dev/null | grep '^{' >> $R; }
BS="16384 32768 65536"
run "register accumulators + atomics (MF_BWD_WIN=0)" MF_BWD_WIN=0
run "LDS window (MF_BWD_WIN=1)" MF_BWD_WIN=1
run "MF_BWD_WIN=0, atomics compiled out (wrong results; time only)" MF_BWD_WIN=0 MONOFORCE_HIP_LIB=$PWD/gpurun_in_ab/xs_noatomic/libmonoforce_hip.so
BS="4096 8192"
run "below one wave per SIMD, one point per lane forced: MF_BWD_WIN=0" MF_BWD_WIN=0 MF_CP_BWD_MAX_WAVES=0 MF_BWD_XS_MIN_WAVES=1
run "below one wave per SIMD, one point per lane forced: MF_BWD_WIN=1" MF_BWD_WIN=1 MF_CP_BWD_MAX_WAVES=0 MF_BWD_XS_MIN_WAVES=1
run "product route (component-parallel)" MF_BWD_WIN=1
cat $R
