#!/bin/bash
# Same-box A/B of several builds of the library (boxes differ by +-5 %, so numbers from different gpurun calls do not
# compare).  Builds are selected with MSDA_LIB (devis_amd/build.py: the in-tree library is never overwritten):
#   python -m devis_amd.build -DMSDA_SOMETHING=1 --out=devis_amd/libmsda_exp_x.so        (in the build container)
#   gpurun -- bash scripts/ab_bench.sh devis_amd/libmsda_hip.so devis_amd/libmsda_exp_x.so [-- bench.py args]
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
    tag=$(basename "$lib" .so)
    MSDA_LIB="$PWD/$lib" python3 bench.py --no-other-configs --no-cpu-baseline --steps 40 "${args[@]}" > "gpurun_out/ab_$tag.json" 2> "gpurun_out/ab_$tag.err" \
      || { echo "$tag: bench failed"; tail -3 "gpurun_out/ab_$tag.err"; continue; }
    python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.loads(open('gpurun_out/ab_%s.json' % tag).read().strip().splitlines()[-1])
print("%-28s %7.3f M-q/s  " % (tag, d['value']) + "  ".join("%s %.4f" % (k.split(' ')[0].replace('msda_', '')[:14], v['avg_ms']) for k, v in d['kernels'].items()), flush=True)
PY
  done
done
