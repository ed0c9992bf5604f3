#!/bin/bash
# Copies the files tools/refresh_evidence.sh TAG left in gpurun_out/ into profiles/ under their committed names.
set -euo pipefail
TAG=$1
G=gpurun_out
P=profiles
cp $G/${TAG}_bench_c2_1080p.json $P/r01_final_bench_c2_1080p.json
cp $G/${TAG}_bench_c2_1080p.err $P/r01_final_breakdown_c2_1080p.txt
cp $G/${TAG}_kernel_stats_c2_1080p.csv $P/r01_final_bench_c2_1080p_kernel_stats.csv
cp $G/${TAG}_bench_c2_1080p_under_rocprof.json $P/r01_final_bench_c2_1080p_line_under_rocprof.json
cp $G/${TAG}_pmc_c2_1080p.json $P/r01_pmc_c2_1080p.json
cp $G/${TAG}_pmc_c2_1080p.md $P/r01_pmc_c2_1080p.md
cp $G/${TAG}_other_workloads.txt $P/r01_final_other_workloads.txt
cp $G/${TAG}_scaffold_c2_kernel_stats.csv $P/r01_scaffold_c2_kernel_stats.csv
ls -la $P
