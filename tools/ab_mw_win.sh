# Bodies of 5..64 points: the multi-wave backward with the workgroup's LDS gradient window (variant build / MF_BWD_WIN) against atomics.
# gpurun -- bash tools/ab_mw_win.sh <other libmonoforce_hip.so>
for lib in $PWD/monoforce_amd/csrc/libmonoforce_hip.so $1; do
  echo "# $lib"
  MONOFORCE_HIP_LIB=$lib python bench.py --workload n32 --no-cpu-baseline --no-others --steps 10 --warmup 2 --detail /tmp/ab_mw_win.json >/dev/null 2>&1; python -c "
import sys,json
d=json.load(open('/tmp/ab_mw_win.json')); print('n32', round(d['ms_per_step'],4), {k: round(v['ms'],4) for k,v in d['roofline']['per_kernel'].items()}, d['config']['launch']['kernels'].get('rollout_bwd_kernel','')[:90])"
  for B in 256 1024 2048; do MONOFORCE_HIP_LIB=$lib AB_B=$B AB_N=8,16,32,64 timeout 300 python tools/ab_points.py 2>/dev/null | grep states; done
done
