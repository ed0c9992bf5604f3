#!/bin/bash
# loss tests + rocprofv3 kernel stats of the scaffold-mode bench, loss/optimizer/neural kernels only (run through gpurun)
set -eo pipefail
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_loss_reference.py tests/test_trainer_gpu.py -m gpu -x -q 2>&1 | tail -2
rm -rf gpurun_out/ssim_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ssim_prof -o run -- python3 bench.py --mode scaffold --workload ${1:-c2} --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ssim_bench.json 2> gpurun_out/ssim_prof.log
python3 - <<'PY'
import csv, json
d = json.loads(open("gpurun_out/ssim_bench.json").read()); print(d["value"], d["ms_per_step"])
for r in csv.DictReader(open("gpurun_out/ssim_prof/run_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("ssim", "adam", "neural", "finish_loss")):
        print(r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3)
PY
