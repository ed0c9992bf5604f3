#!/bin/bash
# Same-box A/B of the in-tree library against the library built from another commit's csrc/ (into a temporary directory, loaded
# through SEGS_RASTER_LIB; nothing in the tree is touched).  usage (on the GPU box): tools/ab_lib.sh /path/to/other/csrc [bench args]
# The caller exports the other tree first, e.g.  mkdir _ab_base && git archive HEAD segs-slam_amd/csrc include | tar -x -C _ab_base   (git-ignored; gpurun_out/ does not travel)
set -eo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OTHER=$1; shift
TMP=$(mktemp -d /tmp/segs_ablib.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
(cd "$ROOT/segs-slam_amd/csrc" && make -s)
cp -r "$OTHER" "$TMP/tree"
(cd "$TMP/tree/segs-slam_amd/csrc" && make -s)
line() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['raster']['kernel_ms']
print('$1', round(d['value'],1), round(d['ms_per_step'],4), {a: k[a] for a in sorted(k)})"; }
for i in 1 2; do
  (cd "$ROOT" && python3 bench.py --no-extras --no-cpu-baseline --breakdown "$@" 2>/dev/null | line "tree ")
  (cd "$ROOT" && SEGS_RASTER_LIB="$TMP/tree/segs-slam_amd/csrc/libsegs_raster.so" python3 bench.py --no-extras --no-cpu-baseline --breakdown "$@" 2>/dev/null | line "other")
done
