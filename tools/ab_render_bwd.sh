#!/bin/bash
# Same-box A/B of render_bwd_kernel's Gaussian role: all-VALU form (default) against the matrix-pipe form (SEGS_RENDER_BWD_MFMA=1).
# usage (GPU box): tools/ab_render_bwd.sh "workload ..." [rounds]
set -eo pipefail
line() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['raster']['kernel_ms']
print('$1', 'it/s', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'bwd', k.get('render_bwd_kernel'), 'fwd', k.get('render_fwd_kernel'))"; }
for wl in $1; do
  echo "== $wl"
  for i in $(seq 1 ${2:-2}); do
    SEGS_RENDER_BWD_MFMA=1 python3 bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | line "mfma"
    python3 bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | line "valu"
  done
done
