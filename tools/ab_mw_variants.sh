mkdir -p gpurun_out/s2
for v in base noatom nogather noex noall; do
  L=""; [ $v != base ] && L=$PWD/gpurun_in_ab/$v/libmonoforce_hip.so
  echo "== $v"
  MONOFORCE_HIP_LIB=$L AB_B=64 AB_N=223 timeout 200 python tools/ab_points.py 2>/dev/null | grep states
  MONOFORCE_HIP_LIB=$L AB_B=1024 AB_N=32 timeout 200 python tools/ab_points.py 2>/dev/null | grep states
done
