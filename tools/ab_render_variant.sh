#!/bin/bash
# Measurement only: build render.o with extra -D flags into a temporary directory and time it through SEGS_RASTER_LIB against
# the in-tree library on the same box.  usage (GPU box): tools/ab_render_variant.sh "workload ..." "ENVVAR=1" -DFLAG [-DFLAG ...]
set -eo pipefail
WLS=$1; ENVS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/segs_variant.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
cd "$ROOT/segs-slam_amd/csrc"
make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -fno-slp-vectorize "$@" -c render.hip -o "$TMP/render.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$TMP/libsegs_variant.so" "$TMP/render.o" $(ls _obj/*.o | grep -v '/render\.o$')
cd "$ROOT"
line() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['raster']['kernel_ms']
print('$1', 'it/s', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'bwd', k.get('render_bwd_kernel'), 'fwd', k.get('render_fwd_kernel'))"; }
for wl in $WLS; do
  echo "== $wl  variant: $ENVS $*"
  for i in 1 2; do
    python3 bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | line "tree   "
    env $ENVS SEGS_RASTER_LIB="$TMP/libsegs_variant.so" python3 bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | line "variant"
  done
done
