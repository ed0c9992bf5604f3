set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in fused unfused; do
  F=0; [ $v = unfused ] && F=16
  export SEGS_RASTER_EXTRA_FLAGS=$F
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/r3_pmcA_$v -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3_pmcA_$v.log
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/r3_pmcB_$v -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3_pmcB_$v.log
  PMC_KEEP_FIRST_STEP=1 python3 tools/pmc_summary.py gpurun_out/r3_pmc_emit_$v.json gpurun_out/r3_pmcA_$v gpurun_out/r3_pmcB_$v | grep -E "duplicate|emit_count|radix_count" || true
done
