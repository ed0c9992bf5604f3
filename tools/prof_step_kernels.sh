#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command, top kernels printed and the stats CSV kept: tools/prof_step_kernels.sh TAG <python args...>
set -eo pipefail
export TMPDIR=/tmp
TAG=$1; shift
rm -rf gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -o run -- python3 "$@" > gpurun_out/${TAG}_under_rocprof.out 2> gpurun_out/${TAG}_prof.log
find gpurun_out/${TAG}_prof -name '*kernel_stats.csv' -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_prof
python3 - gpurun_out/${TAG}_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:30]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:72]
    print(f"{n:72s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.2f} {r['Percentage']}")
PY
