#!/bin/bash
# Same-box A/B of several builds over the per-kernel probe (scripts/scatter_ab.py), two interleaved repetitions:
#   gpurun -- bash scripts/ab_all.sh devis_amd/libmsda_hip.so devis_amd/libmsda_exp_x.so [-- case ...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
libs=(); args=()
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; args=("$@"); break; fi
  [ -f "$1" ] || { echo "no such library: $1" >&2; exit 1; }
  libs+=("$1"); shift
done
for rep in 1 2; do
  for lib in "${libs[@]}"; do
    MSDA_LIB="$PWD/$lib" timeout 600 python3 scripts/scatter_ab.py "${args[@]}" 2> "gpurun_out/ab_all.err" || { echo "$lib: probe failed"; tail -5 gpurun_out/ab_all.err; }
  done
done
