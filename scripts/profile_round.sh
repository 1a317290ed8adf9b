#!/bin/bash
# One profiling round on the GPU box: bench line, rocprofv3 kernel-trace stats of the same command, and the HBM-traffic
# PMC passes (FETCH_SIZE / WRITE_SIZE / L2 hit-miss, separate passes, never combined with a trace).
# usage: scripts/profile_round.sh <tag> [previous bench.json]    -> gpurun_out/<tag>/{bench.json,kernel_stats.csv,pmc_hbm.txt,compare.txt}
# Ends with scripts/compare_bench.py against the previous file (default: the newest profiles/r*_bench.json): every kernel and
# every other_configs time, exit status 1 when one grew by more than 5 %.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p "$O"
cd "$R"
python3 bench.py > "$O/bench.json" 2> "$O/bench.err"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > "$O/kt.log" 2>&1
f=$(find "$O/kt" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$O/kernel_stats.csv"
rm -rf "$O/kt"
cd "$R"
for grp in hbm sq mem; do
  bash scripts/pmc_passes.sh "$O/pmc" scripts/step_only.py scripts/pmc_groups_$grp.txt
  cp "$O/pmc/summary.txt" "$O/pmc_$grp.txt" 2>/dev/null
  rm -rf "$O/pmc"
done
# profiles/hbm_traffic.json for this very build (bench.py quotes it only while the kernel sources are unchanged)
python3 scripts/make_hbm_traffic.py "$O/pmc_hbm.txt" "profiles/$1_pmc_hbm_traffic.txt" > /dev/null && cp profiles/hbm_traffic.json "$O/hbm_traffic.json"
# profiles/onchip.json (busy fractions of the on-chip units) for this very build, same rule
python3 scripts/make_onchip.py "$O/pmc_sq.txt" "$O/pmc_mem.txt" "profiles/$1_pmc_sq.txt" "profiles/$1_pmc_mem.txt" > /dev/null && cp profiles/onchip.json "$O/onchip.json"
tail -c 400 "$O/bench.json"; echo; head -12 "$O/kernel_stats.csv"; cat "$O/pmc_hbm.txt"
prev=${2:-$(ls -t profiles/r*_bench.json 2>/dev/null | head -1)}
if [ -n "$prev" ] && [ -f "$prev" ]; then
  echo "---- against $prev"
  python3 scripts/compare_bench.py "$prev" "$O/bench.json" > "$O/compare.txt"; rc=$?
  grep -E "SLOWER|grew|no time grew" "$O/compare.txt"
  exit $rc
fi
