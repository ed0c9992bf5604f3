#!/bin/bash
# rocprofv3 kernel stats of the anchor-level mapper step at BASELINE config 5's size (300 k anchors, ScanNet model dimensions,
# 1200x680); run through gpurun from the repo root: tools/prof_scaffold_c5.sh TAG
set -eo pipefail
export TMPDIR=/tmp
TAG=${1:-rXX}
rm -rf gpurun_out/${TAG}_scaf_c5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_scaf_c5 -o run -- python3 bench.py --mode scaffold --workload c2 --anchors 300000 --appearance-dim 16 --no-feat-bank --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_scaffold_c5_line_under_rocprof.json 2> gpurun_out/${TAG}_scaf_c5.log
cp gpurun_out/${TAG}_scaf_c5/run_kernel_stats.csv gpurun_out/${TAG}_scaffold_c5_kernel_stats.csv
rm -rf gpurun_out/${TAG}_scaf_c5
python3 bench.py --mode scaffold --workload c2 --anchors 300000 --appearance-dim 16 --no-feat-bank --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_scaffold_c5_line.json 2>/dev/null
python3 - <<PY
import csv, json
d = json.load(open('gpurun_out/${TAG}_scaffold_c5_line.json')); print("it/s", round(d['value'], 1), "ms/step", round(d['ms_per_step'], 4))
for r in list(csv.DictReader(open('gpurun_out/${TAG}_scaffold_c5_kernel_stats.csv')))[:12]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
    print(f"{n:44s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.2f} {r['Percentage']}")
PY
