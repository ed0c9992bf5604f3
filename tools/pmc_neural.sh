#!/bin/bash
# SQ counters of the neural-Gaussian kernels in the anchor-level step at config 5's size (run through gpurun from the repo root):
#   tools/pmc_neural.sh TAG        -> gpurun_out/TAG_pmc_neural_{pair,one_role}.json
# Each counter set is its own rocprofv3 pass (--kernel-trace --pmc only).
set -eo pipefail
export TMPDIR=/tmp
TAG=${1:-rXX}
OUT=gpurun_out
mkdir -p $OUT
CMD="bench.py --mode scaffold --workload c2 --anchors 300000 --appearance-dim 16 --no-feat-bank --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
      "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES"
      "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"
      "FETCH_SIZE" "WRITE_SIZE")
for mode in pair one_role; do
  # bench.py --neural-flags N: segs_neural_set_flags bits (1 = SEGS_NEURAL_ONE_KERNEL_BACKWARD, segs_neural.h)
  if [ $mode = one_role ]; then NF="--neural-flags 1"; else NF=""; fi
  dirs=""
  for i in "${!SETS[@]}"; do
    d=$OUT/${TAG}_pmcn_${mode}_$i
    rm -rf $d
    rocprofv3 --kernel-trace --pmc ${SETS[$i]} --output-format csv -d $d -o run -- python3 $CMD $NF > /dev/null 2> $OUT/${TAG}_pmcn_${mode}_$i.log || { echo "pass $i ($mode) failed"; tail -5 $OUT/${TAG}_pmcn_${mode}_$i.log; }
    dirs="$dirs $d"
    echo "pass $mode $i done"
  done
  python3 tools/pmc_summary.py $OUT/${TAG}_pmc_neural_${mode}.json $dirs > /dev/null
  rm -rf $dirs
done
python3 - <<PY
import json
for mode in ("pair", "one_role"):
    d = json.load(open("$OUT/${TAG}_pmc_neural_%s.json" % mode))
    for k, e in d.items():
        if "neural" in k:
            print(mode, k, {c: round(v) for c, v in e.items()})
PY
