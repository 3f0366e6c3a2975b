#!/bin/bash
# A/B builds of the record-reading backward (rollout_bwd_mw_kernel.h): the launch without its atomics / map gathers / exchange.
# Build first (on the build host):
#   for v in "noatom -DMF_NO_ATOMICS" "nogather -DMF_MW_DBG_NOGATHER" "noex -DMF_MW_DBG_NOEXCHANGE" \
#            "noall -DMF_NO_ATOMICS -DMF_MW_DBG_NOGATHER -DMF_MW_DBG_NOEXCHANGE"; do set -- $v; n=$1; shift
#     bash tools/build_variant.sh $n "$*" monoforce_amd/csrc/rollout_bwd_mw_fast.hip; done
# then on the GPU box: bash tools/ab_mw_variants.sh
cd "$(dirname "$0")/.."
for v in base noatom nogather noex noall; do
  echo "== $v"
  if [ $v = base ]; then unset MONOFORCE_HIP_LIB; else export MONOFORCE_HIP_LIB=$PWD/gpurun_in_ab/$v/libmonoforce_hip.so; fi
  AB_B=64 AB_N=223 timeout 200 python tools/ab_points.py 2>/dev/null | grep states
  AB_B=1024 AB_N=32 timeout 200 python tools/ab_points.py 2>/dev/null | grep states
done
