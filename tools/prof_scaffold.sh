#!/bin/bash
# rocprofv3 kernel stats of the scaffold-mode bench (run through gpurun from the repo root): tools/prof_scaffold.sh [workload]
set -eo pipefail
export TMPDIR=/tmp
WL=${1:-c2}
rm -rf gpurun_out/prof_scaffold
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_scaffold -o run -- python3 bench.py --mode scaffold --workload $WL --steps 25 --warmup 5 --no-cpu-baseline > gpurun_out/scaffold_${WL}_prof.json 2> gpurun_out/prof_scaffold.log
python3 - <<'PY'
import csv, json, glob
f = glob.glob('gpurun_out/scaffold_*_prof.json')[0]
print("ms_per_step", json.load(open(f))['ms_per_step'])
for r in list(csv.DictReader(open('gpurun_out/prof_scaffold/run_kernel_stats.csv')))[:16]:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0][-40:]
    print(f"{n:40s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.2f} {r['Percentage']}")
PY
