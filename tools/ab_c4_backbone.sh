#!/bin/bash
# BASELINE configs[3] (encoder train step): the backbone's launch diet (MF_BACKBONE_LEAN, monoforce_amd/backbones.py) on / off --
# step time from a plain bench run, launches and kernel time per step from a rocprofv3 kernel trace of the same command.
# gpurun -- bash tools/ab_c4_backbone.sh <tag>      -> gpurun_out/<tag>/<tag>_ab_c4_backbone.txt (+ the lean run's kernel stats)
TAG=${1:-rX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$OUT/${TAG}_ab_c4_backbone.txt
: > $R
run() {      # name, env assignments...
  name=$1; shift
  env "$@" timeout ${AB_TIMEOUT:-300} python bench.py --workload c4 --steps 10 --warmup 4 --no-cpu-baseline > /tmp/c4_$name.json 2> /tmp/c4_$name.err
  rm -rf /tmp/prof_$name      # (traces stay on the box: gpurun_out/ is capped at 64 MiB)
  env "$@" timeout ${AB_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o c4 -- python bench.py --workload c4 --steps 6 --warmup 4 --no-cpu-baseline > /tmp/c4p_$name.json 2> /tmp/c4p_$name.err
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${TAG}_c4_${name}_kernel_stats.csv
  python - "$name" "/tmp/c4_$name.json" "$f" >> $R <<'PY'
import csv, json, sys
name, jf, kf = sys.argv[1:4]
try:
    o = json.loads([l for l in open(jf) if l.startswith('{')][-1])
    ms, launch = o['ms_per_step'], o['config'].get('launch', {}).get('mode')
except Exception as e:
    ms, launch = float('nan'), repr(e)[:60] + ' ' + open(jf.replace('.json', '.err')).read()[-300:]
try:
    rows = list(csv.DictReader(open(kf)))
    steps = max(int(r['Calls']) for r in rows if 'rollout_bwd' in r['Name'])
    tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
    naive = sum(float(r['TotalDurationNs']) for r in rows if 'naive_conv' in r['Name'])
    print(f'{name:8s} {ms:8.3f} ms/step ({launch});  under the trace: {calls / steps:7.1f} launches and {tot / steps / 1e6:6.2f} ms of kernel time per step, naive_conv {100 * naive / tot:4.1f} %')
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
        print(f'      {float(r["TotalDurationNs"]) / steps / 1e3:8.1f} us/step  {int(r["Calls"]) / steps:6.1f} x  {r["Name"][:120]}')
except Exception as e:
    print(f'{name:8s} {ms:8.3f} ms/step ({launch});  no trace: {e!r}')
PY
}
run plain MF_BACKBONE_LEAN=0
run lean MF_BACKBONE_LEAN=1
cat $R
