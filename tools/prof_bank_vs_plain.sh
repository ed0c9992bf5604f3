#!/bin/bash
set -eo pipefail
export TMPDIR=/tmp
for nb in "" "--no-feat-bank"; do
  rm -rf gpurun_out/r4_bank
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_bank -o run -- python3 bench.py --mode scaffold --workload c2 --anchors 200000 $nb --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_bank_line.json 2> gpurun_out/r4_bank.log
  python3 - <<PY
import csv, json
d = json.load(open('gpurun_out/r4_bank_line.json')); print("== feature bank", "off" if "$nb" else "on", "it/s under rocprof", round(d['value'], 1), "ms/step", round(d['ms_per_step'], 4))
for r in list(csv.DictReader(open('gpurun_out/r4_bank/run_kernel_stats.csv')))[:9]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
    print(f"  {n:44s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.2f} {r['Percentage']}")
PY
done
rm -rf gpurun_out/r4_bank
