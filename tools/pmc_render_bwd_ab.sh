#!/bin/bash
# SQ counters of render_bwd_kernel with the Gaussian role on the matrix pipe (SEGS_RENDER_BWD_MFMA=1) and on the VALU (default), same
# box, same workload.  usage (GPU box): tools/pmc_render_bwd_ab.sh TAG [WORKLOAD]  ->  gpurun_out/TAG_pmc_render_bwd_ab.txt
set -eo pipefail
TAG=${1:-rXX}
WL=${2:-1080p_3m}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
RES=$OUT/${TAG}_pmc_render_bwd_ab.txt
: > $RES
for arm in valu mfma; do
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"; do
    d=$OUT/${TAG}_pmcab_$arm
    rm -rf $d
    if [ $arm = mfma ]; then export SEGS_RENDER_BWD_MFMA=1; else unset SEGS_RENDER_BWD_MFMA; fi
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o run -- python3 bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/${TAG}_pmcab_$arm.log
    python3 - "$d" "$arm" >> $RES <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d, arm = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(float)
    for row in csv.DictReader(open(f)):
        if "render_bwd" not in row["Kernel_Name"]:
            continue
        per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (disp, c), v in per.items():
        acc[c][0] += v; acc[c][1] += 1
for c in sorted(acc):
    print(f"{arm:5s} {c:28s} {acc[c][0] / acc[c][1]:16.0f} per launch ({acc[c][1]} launches)")
PY
    rm -rf $d
  done
done
cat $RES
