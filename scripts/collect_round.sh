#!/bin/bash
# Copy what scripts/profile_round.sh <tag> left in gpurun_out/<tag>/ (scratch, merged back from the GPU box) into profiles/
# (tracked) under the names the docs and profiles/hbm_traffic.json refer to.     usage: scripts/collect_round.sh <tag>
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/$1
[ -f "$O/bench.json" ] || { echo "no $O/bench.json" >&2; exit 1; }
cp "$O/bench.json" "profiles/$1_bench.json"
for pair in kernel_stats.csv:bench_kernel_stats.csv pmc_hbm.txt:pmc_hbm_traffic.txt pmc_sq.txt:pmc_sq.txt pmc_mem.txt:pmc_mem.txt compare.txt:compare.txt; do
  [ -f "$O/${pair%%:*}" ] && cp "$O/${pair%%:*}" "profiles/$1_${pair##*:}"
done
[ -f "$O/hbm_traffic.json" ] && cp "$O/hbm_traffic.json" profiles/hbm_traffic.json
[ -f "$O/onchip.json" ] && cp "$O/onchip.json" profiles/onchip.json
ls -la profiles/$1_* profiles/hbm_traffic.json
