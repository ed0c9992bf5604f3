set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for F in 0 64; do
  export SEGS_RASTER_EXTRA_FLAGS=$F
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/r3_pmcF_$F -o run -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> gpurun_out/r3_pmcF_$F.log
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/r3_pmcG_$F -o run -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> gpurun_out/r3_pmcG_$F.log
  python3 tools/pmc_summary.py gpurun_out/r3_pmc_fwd_$F.json gpurun_out/r3_pmcF_$F gpurun_out/r3_pmcG_$F | grep -E "render_fwd" || true
done
