cd $GRAFT_REPO_ROOT
export MSDA_ENABLE_HOOKS=1
bash scripts/ab_bench.sh devis_amd/libmsda_hip.so | head -1
for nt in 1 2 4; do
MSDA_FWD_RS_NT=$nt MSDA_BWD_RS_TPW=$nt bash scripts/ab_bench.sh devis_amd/libmsda_exp_t512.so | head -1
done
