#!/bin/bash
# rocprofv3 kernel stats of the Replica mapper step with the frequency regulariser on, fused path and autograd mirror;
# run through gpurun from the repo root: tools/prof_replica_step.sh TAG
set -eo pipefail
export TMPDIR=/tmp
TAG=${1:-rXX}
for v in fused autograd_mirror; do
  rm -rf gpurun_out/${TAG}_rs_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_rs_$v -o run -- python3 tools/replica_step.py --variant $v > gpurun_out/${TAG}_replica_step_${v}_under_rocprof.json 2> gpurun_out/${TAG}_rs_$v.log
  cp gpurun_out/${TAG}_rs_$v/run_kernel_stats.csv gpurun_out/${TAG}_replica_step_${v}_kernel_stats.csv
  rm -rf gpurun_out/${TAG}_rs_$v
done
python3 tools/replica_step.py --variant fused > gpurun_out/${TAG}_replica_step_fused.json 2>/dev/null
python3 tools/replica_step.py --variant autograd_mirror > gpurun_out/${TAG}_replica_step_autograd_mirror.json 2>/dev/null
python3 - <<PY
import csv, json
for v in ("fused", "autograd_mirror"):
    d = json.load(open('gpurun_out/${TAG}_replica_step_%s.json' % v))[v]
    print(v, "it/s", round(d['iters_per_s'], 1), "ms/step", round(d['ms_per_step'], 4), d['phase_ms'])
    for r in list(csv.DictReader(open('gpurun_out/${TAG}_replica_step_%s_kernel_stats.csv' % v)))[:22]:
        n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
        print(f"  {n:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.2f} {r['Percentage']}")
PY
