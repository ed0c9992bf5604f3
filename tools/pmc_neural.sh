#!/bin/bash
# One SQ counter pass over the config-5-sized anchor-level step (through gpurun): tools/pmc_neural.sh TAG
set -eo pipefail
TAG=${1:-rXX}
export TMPDIR=/tmp
OUT=gpurun_out
rm -rf $OUT/${TAG}_pmcn
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/${TAG}_pmcn -o run -- python3 bench.py --mode scaffold --workload c2 --anchors 300000 --appearance-dim 16 --no-feat-bank --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_pmcn.log
python3 tools/pmc_summary.py $OUT/${TAG}_pmc_neural.json $OUT/${TAG}_pmcn > $OUT/${TAG}_pmc_neural.md
rm -rf $OUT/${TAG}_pmcn
grep "neural" $OUT/${TAG}_pmc_neural.md
