#!/bin/bash
# same-box A/B of the matrix-pipe scatter's build variants (MSDA_LIB), then the static-schedule rotation of the owner kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
export MSDA_ENABLE_HOOKS=1
for lib in libmsda_hip.so libmsda_exp_tb2sb0.so libmsda_exp_tb5sb1.so libmsda_exp_tb1sb1.so libmsda_exp_tb10sb0.so; do
  [ -f devis_amd/$lib ] || continue
  echo "== $lib"
  MSDA_LIB=$PWD/devis_amd/$lib timeout 200 python scripts/mfma_check.py dec16 dec16_bf16 2>&1 | grep -v amdgpu.ids
done
echo "== rotation off (MSDA_SCATTER_DBG=4096)"
MSDA_SCATTER_DBG=4096 timeout 200 python scripts/mfma_check.py decB decS 2>&1 | grep -v amdgpu.ids
echo "== rotation on"
timeout 200 python scripts/mfma_check.py decB decS dec16 2>&1 | grep -v amdgpu.ids
