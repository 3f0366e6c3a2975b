#!/bin/bash
# Arbitrary PMC groups for one bench workload: bash tools/pmc_groups.sh <batch> <workload> "<grp1>" "<grp2>" ...   (one rocprofv3 pass per group)
Bn=$1; W=$2; shift 2
OUT=gpurun_out/pmc_groups_$Bn
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o p -- python bench.py --steps 3 --warmup 1 --workload $W --batch $Bn --no-cpu-baseline --no-others > /dev/null 2> $OUT/g$i.err
  f=$(find $OUT/g$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" rollout | sed 's/^[^{]*//'; else echo "group [$grp] failed: $(grep -i -m2 'error\|invalid\|not' $OUT/g$i.err | head -2)"; fi
done
