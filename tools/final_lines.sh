#!/bin/bash
# The round's final bench lines (through gpurun from the repo root): tools/final_lines.sh TAG
#   gpurun_out/TAG_bench_<workload>.json / .err  : default line and c2_1080p, with --breakdown
#   gpurun_out/TAG_other_workloads.txt           : the other workloads and modes at the same commit
set -eo pipefail
TAG=${1:-rXX}
OUT=gpurun_out
for WL in 1080p_3m c2_1080p; do
  timeout -k 10 400 python3 bench.py --workload $WL --breakdown > $OUT/${TAG}_bench_$WL.json 2> $OUT/${TAG}_bench_$WL.err
  echo "$WL done"
done
O=$OUT/${TAG}_other_workloads.txt
echo "workload  iters/s  ms/step  num_rendered(reference R)  instances_live  dominant kernel  its avg ms  whole-step algorithmic GB/s" > $O
line() { python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']; r = d.get('raster') or {}
print('$1', round(d['value'], 1), round(d['ms_per_step'], 4), c.get('num_rendered', ''), c.get('instances_live', ''), d['roofline']['kernel'] if d.get('roofline') else '', round(d['roofline']['avg_launch_ms'], 4) if d.get('roofline') else '', round(r.get('whole_step_GBps', 0), 1) if r else '')"; }
for WL in c1 c2 1080p_1m 1080p_2m c5; do
  timeout -k 10 300 python3 bench.py --workload $WL --no-cpu-baseline --no-extras 2>/dev/null | line $WL >> $O; echo "$WL done"
done
timeout -k 10 300 python3 bench.py --mode trainer --workload c2_1080p --no-cpu-baseline --no-extras 2>/dev/null | line "trainer_c2_1080p" >> $O
timeout -k 10 300 python3 bench.py --mode trainer --workload c4 --no-cpu-baseline --no-extras 2>/dev/null | line "trainer_c4" >> $O
timeout -k 10 300 python3 bench.py --mode scaffold --workload c2 --no-cpu-baseline --no-extras 2>/dev/null | line "scaffold_c2_50k_anchors" >> $O
timeout -k 10 300 python3 bench.py --mode scaffold --workload c2 --anchors 300000 --appearance-dim 16 --no-feat-bank --no-cpu-baseline --no-extras 2>/dev/null | line "scaffold_config5_(300k_anchors,_app16,_no_bank,_1200x680)" >> $O
timeout -k 10 300 python3 bench.py --workload 1080p_3m --sync-forward --no-cpu-baseline --no-extras 2>/dev/null | line "1080p_3m_reference-shaped_entry_points_(sync_forward)" >> $O
cat $O
