#!/bin/bash
# A/B build of the library: recompiles the named translation units with extra flags and links them with the current objects of
# monoforce_amd/csrc into gpurun_in_ab/<name>/libmonoforce_hip.so (travels to the GPU box; select with MONOFORCE_HIP_LIB=...).
#   tools/build_variant.sh <name> "<extra flags>" <tu.hip> [<tu.hip> ...]
set -e
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/monoforce_amd/csrc
out=$root/gpurun_in_ab/$name
mkdir -p "$out" /tmp/mf_variant_$name
objs=""
for o in "$src"/*.o; do
  b=$(basename "$o" .o); skip=0
  for tu in "$@"; do [ "$b" = "$(basename "$tu" .hip)" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for tu in "$@"; do
  b=$(basename "$tu" .hip)
  contract="-ffp-contract=off"; case "$b" in *_fast|*_cost) contract="-ffp-contract=fast-honor-pragmas";; esac
  extra=""; [ "$b" = rollout_bwd_dyn_cp_fast ] && extra="-mllvm -amdgpu-sched-strategy=max-ilp"
  (cd "$src" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize $contract $extra $flags -Wall -Wno-unused-variable -c "$b.hip" -o /tmp/mf_variant_$name/$b.o 2>&1 | grep -E "error|undefined" ) &
  objs="$objs /tmp/mf_variant_$name/$b.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libmonoforce_hip.so" $objs
echo "$out/libmonoforce_hip.so"
