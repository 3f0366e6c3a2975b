for blk in 64 256; do
echo "== MF_CP_BLOCK=$blk"
MF_CP_BLOCK=$blk bash tools/pmc_cmd.sh "python bench.py --steps 3 --warmup 1 --workload c3 --batch 4096 --no-cpu-baseline --no-others" rollout_bwd "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
done
rm -rf gpurun_out/pmc_cmd
