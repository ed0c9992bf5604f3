#!/bin/bash
# Same-box A/B of the frequency regulariser's transforms: this library's own kernels (real_fft.h, default) against the vendor
# FFT library (SEGS_FREQ_HIPFFT=1), on the Replica mapper step.  usage (GPU box): tools/ab_freq_fft.sh [rounds]
set -eo pipefail
line() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read())['fused']; print('$1', round(d['iters_per_s'],1), 'it/s', round(d['ms_per_step'],4), 'ms  freq_loss', d['phase_ms']['freq_loss'], ' early', round(d['early']['ms_per_step'],4))"; }
for i in $(seq 1 ${1:-2}); do
  python3 tools/replica_step.py --variant fused 2>/dev/null | line "own   "
  SEGS_FREQ_HIPFFT=1 python3 tools/replica_step.py --variant fused 2>/dev/null | line "hipfft"
done
