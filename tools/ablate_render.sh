#!/bin/bash
# Measurement only: build render.o with one ABLATE_* macro at a time INTO A TEMPORARY DIRECTORY and time the two tile kernels
# through SEGS_RASTER_LIB (results are WRONG by construction; this only shows where render_bwd_kernel's time goes).  The
# in-tree libsegs_raster.so and _obj/ are never touched.  usage (on the GPU box): tools/ablate_render.sh [workload]
set -eo pipefail
WL=${1:-c2_1080p}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/segs_ablate.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
cd "$ROOT/segs-slam_amd/csrc"
make -s
for v in BASE ABLATE_NO_ATOMICS ABLATE_NO_GAUSS_ROLE ABLATE_NO_TRANS "ABLATE_NO_GAUSS_ROLE -DABLATE_NO_TRANS"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -fno-slp-vectorize -DSEGS_MEASURE -D$v -c render.hip -o "$TMP/render.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$TMP/libsegs_ablate.so" "$TMP/render.o" $(ls _obj/*.o | grep -v '/render\.o$')
  echo "== $v"
  (cd "$ROOT" && SEGS_RASTER_LIB="$TMP/libsegs_ablate.so" python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['raster']['kernel_ms']; print('ms/step', round(d['ms_per_step'],4), 'bwd', k.get('render_bwd_kernel'), 'fwd', k.get('render_fwd_kernel'))")
done
