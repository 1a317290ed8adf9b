#!/bin/bash
# Build the in-tree library and any number of experimental variants, failing loudly:  scripts/bx.sh [name:-DFLAG[,-DFLAG2] ...]
# -> devis_amd/libmsda_hip.so, devis_amd/libmsda_exp_<name>.so
set -e
cd "$(dirname "$0")/.."
python -m devis_amd.build > /tmp/bx_main.log 2>&1 || { grep -E "error" -A3 /tmp/bx_main.log | head -30; echo "MAIN BUILD FAILED"; exit 1; }
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  python -m devis_amd.build ${flags//,/ } --out=devis_amd/libmsda_exp_$name.so > /tmp/bx_$name.log 2>&1 || { grep -E "error" -A3 /tmp/bx_$name.log | head -30; echo "BUILD OF $name FAILED"; exit 1; }
done
ls -la devis_amd/*.so | awk '{print $5, $9}'
echo BUILD OK
