#!/bin/bash
# Measurement only: the whole library built with another radix-sort tile size (keys per thread of the 256-thread sort workgroups;
# the tree's is 8 = 2048 keys per workgroup) into a temporary directory, timed through SEGS_RASTER_LIB against the tree's on the
# small-problem steps (Replica mapper step, config-4 trainer step).  usage (GPU box): tools/ab_sort_tile.sh "2 4"
set -eo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/segs_sorttile.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
cd "$ROOT/segs-slam_amd/csrc" && make -s
run() {   # $1 = label, env SEGS_RASTER_LIB set by the caller or not
  (cd "$ROOT" && python3 tools/replica_step.py --variant fused 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())['fused']; print('$1 replica', round(d['iters_per_s'],1), 'it/s', round(d['ms_per_step'],4), 'ms', {k: round(v,4) for k,v in d['phase_ms'].items()})")
  (cd "$ROOT" && python3 bench.py --mode trainer --workload c4 --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 trainer c4', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms')")
}
run "tree(8)"
for ipt in $1; do
  mkdir -p $TMP/o$ipt
  make -s OBJ=$TMP/o$ipt LIB=$TMP/lib$ipt.so COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../include -DSEGS_SORT_ITEMS_PER_THREAD=$ipt" -j8
  export SEGS_RASTER_LIB=$TMP/lib$ipt.so
  run "ipt=$ipt "
  unset SEGS_RASTER_LIB
done
run "tree(8)"
