#!/bin/bash
# PMC groups for an arbitrary command: bash tools/pmc_cmd.sh "<command>" <kernel-name filter> "<grp1>" "<grp2>" ...
CMD=$1; FILT=$2; shift 2
OUT=gpurun_out/pmc_cmd
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rm -rf $OUT/g$i
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o p -- $CMD > /dev/null 2> $OUT/g$i.err
  f=$(find $OUT/g$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" $FILT | cut -c1-50,95-; else echo "group [$grp] failed"; fi
done
