#!/bin/bash
# Same-box A/B of two builds of the library (boxes differ by +-5 %, so numbers from different gpurun calls do not compare):
# devis_amd/libmsda_hip.so (built from the committed source, hash file current -> no rebuild on the box) against
# devis_amd/libmsda_exp.so (built by hand from the experimental source:  hipcc ... -o devis_amd/libmsda_exp.so), two
# rounds each.  usage: gpurun -- bash scripts/ab_bench.sh [bench.py args]
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for rep in 1 2; do
  python3 bench.py --no-other-configs --no-cpu-baseline "$@" > gpurun_out/ab_base.json 2>/dev/null
  cp devis_amd/libmsda_hip.so /tmp/keep.so; cp devis_amd/libmsda_exp.so devis_amd/libmsda_hip.so
  python3 bench.py --no-other-configs --no-cpu-baseline "$@" > gpurun_out/ab_exp.json 2>/dev/null
  cp /tmp/keep.so devis_amd/libmsda_hip.so
  python3 -c "
import json
for f in ('ab_base','ab_exp'):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['value'], {k[:18]:v['avg_ms'] for k,v in d['kernels'].items()})"
done
