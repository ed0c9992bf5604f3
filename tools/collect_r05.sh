#!/bin/bash
# Round 5's evidence files beside the headline set (through gpurun from the repo root): tools/collect_r05.sh
#   gpurun_out/r05_replica_step_{own,hipfft}.json + _kernel_stats.csv : the Replica mapper step with the regulariser's transforms by
#                                                                      this library's kernels / by the vendor FFT library
#   gpurun_out/r05_freq_loss_alone.txt                                : the regulariser alone at the three folded sizes, both arms
set -eo pipefail
export TMPDIR=/tmp
for arm in own hipfft; do
  if [ $arm = hipfft ]; then export SEGS_FREQ_HIPFFT=1; else unset SEGS_FREQ_HIPFFT; fi
  rm -rf gpurun_out/r05_rs_$arm
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_rs_$arm -o run -- python3 tools/replica_step.py --variant fused > /dev/null 2> gpurun_out/r05_rs_$arm.log
  find gpurun_out/r05_rs_$arm -name '*kernel_stats.csv' -exec cp {} gpurun_out/r05_replica_step_${arm}_kernel_stats.csv \;
  rm -rf gpurun_out/r05_rs_$arm gpurun_out/r05_rs_$arm.log
  python3 tools/replica_step.py --variant fused > gpurun_out/r05_replica_step_$arm.json 2>/dev/null
done
: > gpurun_out/r05_freq_loss_alone.txt
for sz in "680 1200" "480 640" "1080 1920"; do
  unset SEGS_FREQ_HIPFFT
  python3 tools/time_freq_loss.py $sz 2>/dev/null >> gpurun_out/r05_freq_loss_alone.txt
  SEGS_FREQ_HIPFFT=1 python3 tools/time_freq_loss.py $sz 2>/dev/null >> gpurun_out/r05_freq_loss_alone.txt
done
unset SEGS_FREQ_HIPFFT
cat gpurun_out/r05_freq_loss_alone.txt
python3 - <<'PY'
import json
for arm in ("own", "hipfft"):
    d = json.load(open(f"gpurun_out/r05_replica_step_{arm}.json"))["fused"]
    print(arm, round(d["iters_per_s"], 1), "it/s", round(d["ms_per_step"], 4), "ms", d["phase_ms"], "early", round(d["early"]["ms_per_step"], 4))
PY
