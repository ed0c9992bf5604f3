#!/bin/bash
# raster tests + bench breakdown (run through gpurun): tools/quick_bench.sh [workload]
set -eo pipefail
WL=${1:-c2_1080p}
timeout -k 10 400 python -m pytest tests/test_raster_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 200 python bench.py --workload $WL --no-cpu-baseline --breakdown 2> gpurun_out/qb.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
head -6 gpurun_out/qb.err
