# Backward between the component-parallel range (<= 8192 rollouts) and one wave per SIMD (16 384): general kernels vs the positions-only LDS-window kernels vs early recompute at > 2 waves per SIMD.  gpurun -- bash tools/ab_between.sh
run() { echo "# $1"; shift; env "$@" python tools/ab_mw_small.py 10240 12288 14336 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['B'], {k: round(v,4) for k,v in d['ms'].items() if 'rollout' in k}, d['gz_norm'], d['kernels']['rollout_bwd_kernel'][:95], '| fwd', d['kernels']['rollout_fwd_kernel'][:80])"; }
run default X=1
run "XS from half a wave per SIMD" MF_BWD_XS_MIN_WAVES=512
run "cp early up to 4 waves per SIMD" MF_CP_BWD_MAX_WAVES=4096
