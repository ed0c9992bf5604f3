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
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD"; do
  # (a fifth set -- TA_* / TCP_* stall counters -- is refused on this pool's boxes ("exceeds the capabilities of the hardware")
  # and the aborted profiler then sits until the run is killed: left out)
  i=$((i+1))
  rm -rf $OUT/${TAG}_pmcs_$i
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${TAG}_pmcs_$i -o run -- $CMD > /dev/null 2> $OUT/${TAG}_pmcs_$i.log
  echo "pass $i done"
  DIRS="$DIRS $OUT/${TAG}_pmcs_$i"
done
python3 tools/pmc_summary.py $OUT/${TAG}_pmc_scaffold_c5.json $DIRS > $OUT/${TAG}_pmc_scaffold_c5.md
for d in $DIRS; do rm -rf $d; done
echo "all done"
