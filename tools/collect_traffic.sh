#!/bin/bash
# The PMC-traffic part of tools/collect_profiles.sh alone (after a library change that leaves the side sweeps alone): default line, kernel trace of
# c3 / c3f, FETCH / WRITE passes, hbm_traffic.json.   gpurun -- bash tools/collect_traffic.sh <tag>
TAG=${1:-rX}
OUT=gpurun_out/$TAG
mkdir -p $OUT /tmp/mfprof
export TMPDIR=/tmp
B="--no-cpu-baseline --no-others"
timeout 900 python bench.py --detail $OUT/${TAG}_bench_default_detail.json > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err      # (stdout: the compact line the driver parses; --detail: tables, sweep, side workloads)
for w in c3f c3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mfprof/prof_$w -o $w -- python bench.py --steps 20 --warmup 3 --workload $w $B --detail $OUT/${TAG}_${w}_bench_under_rocprof_detail.json > $OUT/${TAG}_${w}_bench_under_rocprof.json 2> $OUT/prof_$w.err
  f=$(find /tmp/mfprof/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -40 "$f" > $OUT/${TAG}_${w}_kernel_stats.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  for w in c3f c3; do      # (c4's PMC passes -- ~1000 dispatches per step, serialised by the counters -- run into the timeout: no c4 row)
    timeout 240 rocprofv3 --pmc $c --output-format csv -d /tmp/mfprof/pmc_${c}_$w -o $w -- python bench.py --steps 4 --warmup 1 --workload $w $B --detail /tmp/mfprof/d.json > /dev/null 2> $OUT/pmc_${c}_$w.err
    f=$(find /tmp/mfprof/pmc_${c}_$w -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" > $OUT/${TAG}_pmc_${c}_$w.txt
  done
done
# ... and the saturated batches of the sweep (`batch_sweep` rows carry frac_traffic at 8192 / 16 384 / 32 768): the c3 step and
# the plain forward (c3f) at those batches, same two counters
for Bn in ${MF_SWEEP_PMC:-8192 16384 32768}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    for w in c3f c3; do
      timeout 240 rocprofv3 --pmc $c --output-format csv -d /tmp/mfprof/pmc_${c}_${w}_B$Bn -o $w -- python bench.py --steps 3 --warmup 1 --workload $w --batch $Bn $B --detail /tmp/mfprof/d.json > /dev/null 2> $OUT/pmc_${c}_${w}_B$Bn.err
      f=$(find /tmp/mfprof/pmc_${c}_${w}_B$Bn -name "*counter_collection.csv" | head -1)
      [ -n "$f" ] && python tools/pmc_summary.py "$f" > $OUT/${TAG}_pmc_${c}_${w}_B$Bn.txt
    done
  done
done
# HBM bytes per launch of the hand-written kernels for bench.py's `roofline.traffic`: WRITE_SIZE + 2 x FETCH_SIZE (KiB; on
# gfx950 rocprofv3 tallies a 128-byte read request as 64 bytes -- MI355X_MICROARCH.md, "HBM")
python - "$OUT" "$TAG" <<'PY'
import ast, json, re, sys
out, tag = sys.argv[1], sys.argv[2]
names = {'rollout_fwd': 'rollout_fwd_kernel', 'rollout_bwd': 'rollout_bwd_kernel', 'lift_splat_fwd': 'lift_splat_fwd_kernel',
         'lift_splat_bwd': 'lift_splat_bwd_kernel'}
res = {}
import os
for w in ['c3f', 'c3'] + [f'{w}_B{b}' for b in os.environ.get('MF_SWEEP_PMC', '8192 16384 32768').split() for w in ('c3f', 'c3')]:
    per, calls = {}, {}
    for c, mult in (('FETCH_SIZE', 2), ('WRITE_SIZE', 1)):
        try:
            best = {}       # per kernel family: the template instantiation with the most dispatches (the timed steps', not the set-up rollout's)
            for line in open(f'{out}/{tag}_pmc_{c}_{w}.txt'):
                m = re.match(r'(.*?) (\{.*?\})(?: dispatches (\d+))?\s*$', line)
                if not m:
                    continue
                key = next((v for k, v in names.items() if k in m.group(1)), None)
                n = int(m.group(3) or 1)
                if key and n > best.get(key, (0, 0))[0]:
                    best[key] = (n, mult * ast.literal_eval(m.group(2))[c] * 1024)
            for key, (n, val) in best.items():
                per[key] = per.get(key, 0) + val
        except FileNotFoundError:
            pass
    if per:
        res[w] = per
import hashlib
res['_library_sha256'] = hashlib.sha256(open('monoforce_amd/csrc/libmonoforce_hip.so', 'rb').read()).hexdigest()      # bench.py refuses the file for any other build
json.dump(res, open(f'{out}/hbm_traffic.json', 'w'), indent=1)
print(json.dumps(res))
PY
cp $OUT/hbm_traffic.json profiles/hbm_traffic.json
timeout 400 python bench.py --detail $OUT/${TAG}_bench_default_with_traffic_detail.json > $OUT/${TAG}_bench_default_with_traffic.json 2> /dev/null
ls $OUT
