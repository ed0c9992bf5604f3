#!/bin/bash
# usage (GPU box): tools/quick_bench.sh "workload ..."   -- ms per step and the per-kernel table of each workload
for wl in $1; do echo "== $wl"; timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['raster']['kernel_ms'])" || exit 1; done
