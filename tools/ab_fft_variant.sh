#!/bin/bash
# Measurement only: freq_loss.o built with extra -D flags into a temporary directory; per-kernel times of the regulariser alone under
# rocprofv3 for the tree's library and the variant's.  usage (GPU box): tools/ab_fft_variant.sh -DFLAG [-DFLAG ...]
set -eo pipefail
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/segs_fftvar.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
cd "$ROOT/segs-slam_amd/csrc"
make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -ffp-contract=off "$@" -c freq_loss.hip -o "$TMP/freq_loss.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$TMP/libv.so" "$TMP/freq_loss.o" $(ls _obj/*.o | grep -v '/freq_loss\.o$')
cd "$ROOT"
show() { python3 - "$1" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rfft" in r["Name"] or "freq" in r["Name"] or "fft" in r["Name"] or "transpose" in r["Name"] or "add_kernel" in r["Name"]:
        print(f"   {r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.2f}")
PY
}
for arm in tree variant; do
  if [ $arm = variant ]; then export SEGS_RASTER_LIB="$TMP/libv.so"; else unset SEGS_RASTER_LIB; fi
  rm -rf $TMP/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d $TMP/prof -o run -- python3 tools/time_freq_loss.py > $TMP/out.txt 2> $TMP/log.txt
  echo "== $arm $* : $(cat $TMP/out.txt)"
  show $(find $TMP/prof -name '*kernel_stats.csv')
done
