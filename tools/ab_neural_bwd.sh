#!/bin/bash
# Measurement only: the anchor-level mapper step at config 5's size with variants of neural.o built INTO A TEMPORARY DIRECTORY
# (loaded through SEGS_RASTER_LIB; the in-tree library is never touched) and with the one-kernel backward
# (SEGS_NEURAL_ONE_KERNEL_BACKWARD, segs_neural.h).  usage (on the GPU box): tools/ab_neural_bwd.sh ["-DFLAG=.." ...]
set -eo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/segs_ab.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
cd "$ROOT/segs-slam_amd/csrc"
make -s
run() {
  (cd "$ROOT" && python - <<'PY'
import json, torch, bench
import os
from segs_slam_amd import _capi
_capi.lib().segs_neural_set_flags(int(os.environ.get("ONE_KERNEL_BACKWARD", "0")))   # SEGS_NEURAL_ONE_KERNEL_BACKWARD (segs_neural.h)
r = bench.mapper_step_block(torch.device("cuda:0"), steps=40, warmup=10)
print("  ms/step", round(r["ms_per_step"], 4), "p50", round(r["step_ms"]["p50"], 4), {k: v for k, v in r["phase_ms"].items() if "neural" in k})
PY
  )
}
echo "== in-tree (chain + wgrad waves)"; run
echo "== in-tree, SEGS_NEURAL_ONE_KERNEL_BACKWARD"; ONE_KERNEL_BACKWARD=1 run
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -fno-slp-vectorize $v -c neural.hip -o "$TMP/neural.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$TMP/libsegs_ab.so" "$TMP/neural.o" $(ls _obj/*.o | grep -v '/neural\.o$')
  echo "== $v"; SEGS_RASTER_LIB="$TMP/libsegs_ab.so" run
done
