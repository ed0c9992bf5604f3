#!/bin/bash
# usage (GPU box): tools/quick_scaffold.sh   -- config-5-sized and config-3-sized anchor-level steps: it/s, ms, top kernels by HIP events
for args in "--anchors 300000 --appearance-dim 16 --no-feat-bank" "--anchors 50000"; do
  echo "== scaffold c2 $args"
  timeout -k 10 200 python bench.py --mode scaffold --workload c2 $args --steps 30 --warmup 5 --no-cpu-baseline --breakdown 2>gpurun_out/qs.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms')" || exit 1
  grep "ms/step" gpurun_out/qs.err | sort -k2 -n -r | head -8
done
