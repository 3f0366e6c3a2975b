#!/bin/bash
# Kernel-trace durations of the splat kernels at config-4 shapes (min / mean over tools/bench_lift_splat.py's launches; B = 1 and B = 8 rows mixed:
# min = B = 1, max = B = 8).   gpurun -- bash tools/prof_splat.sh <tag>
TAG=${1:-rX}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ls -o ls -- python tools/bench_lift_splat.py > $OUT/${TAG}_bench_lift_splat_under_rocprof.txt 2> /tmp/prof_ls.err
f=$(find /tmp/prof_ls -name "*kernel_stats.csv" | head -1)
python - "$f" > $OUT/${TAG}_lift_splat_kernel_stats.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('splat', 'lift')):
        print(f"{r['Name'].split('(')[0][-60:]:60s} calls {int(r['Calls']):4d}  min {float(r['MinNs']) / 1e3:7.1f} us (B = 1)  max {float(r['MaxNs']) / 1e3:7.1f} us (B = 8)  mean {float(r['AverageNs']) / 1e3:7.1f}")
PY
cat $OUT/${TAG}_lift_splat_kernel_stats.txt
