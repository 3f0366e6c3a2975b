#!/bin/bash
# Times the rollout kernels (tools/ab_cp.py, component-parallel rows) with the product library and with every A/B build under
# gpurun_in_ab/ (tools/build_variant.sh):   AB_B=256,1024 AB_BWD=1 tools/ab_variants.sh [variant ...]
cd "$(dirname "$0")/.."
vs="$@"; [ -z "$vs" ] && vs=$(ls gpurun_in_ab 2>/dev/null)
echo "== product"; python tools/ab_cp.py 2>&1 | grep cp16 | grep states
for v in $vs; do
  echo "== $v"; MONOFORCE_HIP_LIB=$PWD/gpurun_in_ab/$v/libmonoforce_hip.so python tools/ab_cp.py 2>&1 | grep cp16 | grep states
done
