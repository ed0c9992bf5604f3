#!/bin/bash
# per-kernel SQ counters of the scaffold-mode bench: tools/pmc_scaffold.sh "COUNTER ..." [workload]
set -eo pipefail
export TMPDIR=/tmp
WL=${2:-c2}
rm -rf gpurun_out/pmc_scaffold
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d gpurun_out/pmc_scaffold -o run -- python3 bench.py --mode scaffold --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_scaffold.log
python3 tools/pmc_summary.py gpurun_out/pmc_scaffold.json gpurun_out/pmc_scaffold | grep -E "neural|wgrad|render_bwd"
