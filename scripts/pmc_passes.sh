#!/bin/bash
# rocprofv3 PMC passes over a probe script: one counter group per pass (never together with a trace), each
# pass under its own timeout (a counter set the hardware cannot schedule aborts the run and can hang).
# usage: scripts/pmc_passes.sh <out_dir> <script.py> <groups_file>      (env as the probe reads it)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$1; S=$2; GF=$3
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d "$O/p$i" -- python3 "$R/$S" > "$O/p$i.log" 2>&1
  f=$(find "$O/p$i" -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 "$R/scripts/pmc_dump.py" "$f" >> "$O/summary.txt"; else echo "pass $i ($grp): no output" >> "$O/summary.txt"; tail -3 "$O/p$i.log" >> "$O/summary.txt"; fi
done < "$R/$GF"
find "$O" -type f ! -name "*.txt" ! -name "*.log" -delete
