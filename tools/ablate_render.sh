#!/bin/bash
# Measurement only: rebuild render.o with one ABLATE_* macro at a time and time the two tile kernels (results are WRONG by
# construction; this only shows where render_bwd_kernel's time goes).  usage (on the GPU box): tools/ablate_render.sh [workload]
set -eo pipefail
WL=${1:-c2_1080p}
cd segs-slam_amd/csrc
cp libsegs_raster.so /tmp/libsegs_raster.keep
for v in BASE ABLATE_NO_ATOMICS ABLATE_NO_GAUSS_ROLE ABLATE_NO_TRANS "ABLATE_NO_GAUSS_ROLE -DABLATE_NO_TRANS"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -fno-slp-vectorize -D$v -c render.hip -o _obj/render.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsegs_raster.so _obj/*.o
  echo "== $v"
  (cd ../.. && python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['raster']['kernel_ms']; print('ms/step', round(d['ms_per_step'],4), 'bwd', k.get('render_bwd_kernel'), 'fwd', k.get('render_fwd_kernel'))")
done
cp /tmp/libsegs_raster.keep libsegs_raster.so
touch render.hip
