export TMPDIR=/tmp
OUT=gpurun_out/pmc_stream; mkdir -p $OUT
for Bn in 256 1024; do
for c in "SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  n=$(echo $c | tr ' ' '_')
  MF_BENCH_NO_GRAPH=1 timeout 200 rocprofv3 --pmc $c --output-format csv -d $OUT/p_${n}_$Bn -o x -- python bench.py --steps 4 --warmup 1 --workload c3 --batch $Bn --no-cpu-baseline --no-others > /dev/null 2> $OUT/err_$n.txt
  f=$(find $OUT/p_${n}_$Bn -name "*counter_collection.csv" | head -1)
  echo "B=$Bn $c"; python tools/pmc_summary.py "$f" rollout_bwd
done; done
