#!/bin/bash
# PMC passes over the anchor-level mapper step at BASELINE config 5's size (run through gpurun from the repo root):
#   tools/pmc_scaffold_c5.sh TAG      -> gpurun_out/TAG_pmc_scaffold_c5.{json,md}
set -eo pipefail
TAG=${1:-rXX}
export TMPDIR=/tmp
OUT=gpurun_out
CMD="python3 bench.py --mode scaffold --workload c2 --anchors 300000 --appearance-dim 16 --no-feat-bank --steps 4 --warmup 2 --no-cpu-baseline"
i=0
DIRS=""
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" \
  "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rm -rf $OUT/${TAG}_pmcs_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${TAG}_pmcs_$i -o run -- $CMD > /dev/null 2> $OUT/${TAG}_pmcs_$i.log
  echo "pass $i done"
  DIRS="$DIRS $OUT/${TAG}_pmcs_$i"
done
python3 tools/pmc_summary.py $OUT/${TAG}_pmc_scaffold_c5.json $DIRS > $OUT/${TAG}_pmc_scaffold_c5.md
for d in $DIRS; do rm -rf $d; done
echo "all done"
