#!/bin/bash
# Collect the measurement artefacts kept under profiles/ (run on the GPU box: `gpurun -- bash tools/collect_profiles.sh <tag>`).
# Every step has its own timeout; PMC passes are separate from the kernel-trace passes (and from each other).
TAG=${1:-rX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-others"
timeout 600 python bench.py --sweep > $OUT/${TAG}_bench_default_with_sweep.json 2> $OUT/bench_default.err
for w in c3f c3 c4; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- python bench.py --steps 20 --warmup 3 --workload $w $B > $OUT/${TAG}_${w}_bench_under_rocprof.json 2> $OUT/prof_$w.err
  f=$(find $OUT/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -40 "$f" > $OUT/${TAG}_${w}_kernel_stats.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  for w in c3f c3; do
    timeout 240 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${c}_$w -o $w -- python bench.py --steps 4 --warmup 1 --workload $w $B > /dev/null 2> $OUT/pmc_${c}_$w.err
    f=$(find $OUT/pmc_${c}_$w -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" rollout > $OUT/${TAG}_pmc_${c}_$w.txt
  done
done
timeout 600 python tools/bench_kernels.py > $OUT/${TAG}_bench_kernels.jsonl 2> $OUT/bench_kernels.err
AB_B=1024,16384,65536 timeout 300 python tools/bench_planner.py > $OUT/${TAG}_bench_planner.jsonl 2> $OUT/bench_planner.err
timeout 300 python tools/bench_lift_splat.py 2> $OUT/bench_lift_splat.err | grep '^B=' > $OUT/${TAG}_bench_lift_splat.txt
timeout 300 python tools/bench_graphed.py 2> $OUT/bench_graphed.err | grep n_trajs > $OUT/${TAG}_bench_graphed.txt
AB_B=4 AB_N=100,175,223,400 timeout 300 python tools/ab_points.py 2> /dev/null | grep forces > $OUT/${TAG}_large_body_small_batch.txt
AB_B=64 AB_N=100,175,223,400 timeout 300 python tools/ab_points.py 2> /dev/null | grep forces >> $OUT/${TAG}_large_body_small_batch.txt
ls -la $OUT | head -40
