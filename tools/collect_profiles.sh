#!/bin/bash
# Collect the evidence files for profiles/ on the MI355X box (run through gpurun from the repo root):
#   tools/collect_profiles.sh TAG [WORKLOAD]
# Writes gpurun_out/TAG_*: the plain bench line, the rocprofv3 --kernel-trace --stats summary of the same command (and the
# bench line printed under the profiler), and three separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ counters).
set -eo pipefail
TAG=${1:-rXX}
WL=${2:-c2_1080p}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
python3 bench.py --workload $WL --breakdown > $OUT/${TAG}_bench_${WL}.json 2> $OUT/${TAG}_bench_${WL}.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o run -- python3 bench.py --workload $WL --steps 25 --warmup 5 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_${WL}_under_rocprof.json 2> $OUT/${TAG}_prof.log
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -o run -- python3 bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/${TAG}_pmc_$c.log
  echo "pmc $c done"
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_SQ -o run -- python3 bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/${TAG}_pmc_SQ.log
echo "pmc SQ done"
python3 tools/pmc_summary.py $OUT/${TAG}_pmc_${WL}.json $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_SQ > $OUT/${TAG}_pmc_${WL}.md
find $OUT/${TAG}_prof -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_kernel_stats_${WL}.csv \;
# only the summaries travel back (gpurun merges at most 64 MiB)
rm -rf $OUT/${TAG}_prof $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_SQ
echo "all done"
