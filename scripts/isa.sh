#!/bin/bash
# Device assembly + resource usage of one translation unit: scripts/isa.sh msda_rs [-DFLAG ...]  ->  /tmp/isa/<unit>.s, /tmp/isa/<unit>.usage
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
u=$1; shift
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -ffp-contract=off -Wno-pass-failed "$@" \
  -I "$R/include" -I "$R/devis_amd/csrc" --cuda-device-only -S -Rpass-analysis=kernel-resource-usage \
  "$R/devis_amd/csrc/$u.hip" -o /tmp/isa/$u.s 2> /tmp/isa/$u.usage.raw || { cat /tmp/isa/$u.usage.raw | grep -v remark | head -30; exit 1; }
python3 - /tmp/isa/$u.usage.raw > /tmp/isa/$u.usage <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for blk in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = blk.split("\n")[0].strip()
    g = lambda k: (re.search(k + r": (\S+)", blk) or [None, "?"])[1]
    print("%-90s VGPR %s AGPR %s SGPR %s spillV %s spillS %s scratch %s occ %s LDS %s" % (
        name[:90], g("VGPRs"), g("AGPRs"), g("SGPRs"), g("VGPRs Spill"), g("SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
PY
cat /tmp/isa/$u.usage
