#!/bin/bash
# usage (GPU box): tools/quick_ab.sh "workload ..." [ENVVAR]   -- bench lines with and without ENVVAR=1 (default SEGS_SORT_LEGACY)
VAR=${2:-SEGS_SORT_LEGACY}
for wl in $1; do for leg in 0 1; do if [ $leg = 1 ]; then export $VAR=1; else unset $VAR; fi; echo "== $wl $VAR=$leg"; timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['raster']['kernel_ms'])"; done; done
