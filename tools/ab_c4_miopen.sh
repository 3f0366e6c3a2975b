#!/bin/bash
# BASELINE configs[3] (encoder train step) under MIOpen / layout settings: which solvers run, what the step costs.
# gpurun -- bash tools/ab_c4_miopen.sh <tag>      -> gpurun_out/<tag>/<tag>_ab_c4_miopen.txt
TAG=${1:-rX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$OUT/${TAG}_ab_c4_miopen.txt
: > $R
run() {      # name, env assignments...
  name=$1; shift
  rm -rf /tmp/prof_$name      # (traces stay on the box: gpurun_out/ is capped at 64 MiB)
  env "$@" timeout ${AB_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o c4 -- python bench.py --workload c4 --steps 6 --warmup 4 --no-cpu-baseline > /tmp/c4_$name.json 2> /tmp/c4_$name.err
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  python - "$name" "/tmp/c4_$name.json" "$f" >> $R <<'PY'
import csv, json, sys
name, jf, kf = sys.argv[1:4]
try:
    line = [l for l in open(jf) if l.startswith('{')][-1]
    o = json.loads(line)
    ms, launch = o['ms_per_step'], o['config'].get('launch', {}).get('mode')
except Exception as e:
    ms, launch = float('nan'), repr(e)[:60]
naive = tot = calls = 0.0
top = []
try:
    rows = list(csv.DictReader(open(kf)))
    for r in rows:
        t = float(r['TotalDurationNs']); tot += t; calls += int(r['Calls'])
        if 'naive_conv' in r['Name']:
            naive += t
    top = sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:6]
except Exception as e:
    top = []
print(f'{name:28s} {ms:8.3f} ms/step ({launch})  kernel time {tot / 1e6:9.1f} ms over {int(calls)} launches in the run, naive_conv {100 * naive / max(tot, 1):5.1f} %')
for r in top:
    print(f'      {float(r["Percentage"]):5.1f} %  {int(r["Calls"]):6d} x  {r["Name"][:110]}')
PY
}
run base
run benchmark MF_MIOPEN_BENCHMARK=1
run channels_last MF_MIOPEN_BENCHMARK=1 MF_CHANNELS_LAST=1
[ -n "$AB_MORE" ] && run find_normal MF_MIOPEN_BENCHMARK=1 MIOPEN_FIND_MODE=NORMAL
cat $R
