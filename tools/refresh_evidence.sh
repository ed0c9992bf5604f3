#!/bin/bash
# Regenerates every evidence file of profiles/ on the MI355X box (run through gpurun from the repo root):
#   tools/refresh_evidence.sh TAG            -> gpurun_out/TAG_*   (then tools/install_evidence.sh TAG copies them into profiles/)
set -eo pipefail
TAG=${1:-rXX}
tools/collect_profiles.sh $TAG c2_1080p
OUT=gpurun_out/${TAG}_other_workloads.txt
: > $OUT
for WL in c1 c2 1080p_1m 1080p_2m 1080p_3m c5; do
  timeout -k 10 300 python3 bench.py --workload $WL --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$WL', round(d['value'], 1), round(d['ms_per_step'], 4), d['config']['num_rendered'], round(d['roofline']['avg_launch_ms'], 4), round(d['raster']['achieved_GBps'], 1))" >> $OUT
  echo "$WL done"
done
for M in "trainer c2_1080p" "trainer c4" "scaffold c2_1080p" "scaffold c2"; do
  set -- $M
  timeout -k 10 300 python3 bench.py --mode $1 --workload $2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', '$2', round(d['value'], 1), round(d['ms_per_step'], 4), d['config']['num_rendered'])" >> $OUT
  echo "$M done"
done
timeout -k 10 300 python3 bench.py --mode scaffold --workload c5 --anchors 300000 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('scaffold c5 300k-anchors', round(d['value'], 1), round(d['ms_per_step'], 4), d['config']['num_rendered'])" >> $OUT
tools/prof_scaffold.sh c2 > gpurun_out/${TAG}_scaffold_c2_top.txt
cp gpurun_out/prof_scaffold/run_kernel_stats.csv gpurun_out/${TAG}_scaffold_c2_kernel_stats.csv
cat $OUT
