cd $GRAFT_REPO_ROOT
bash scripts/ab_bench.sh devis_amd/libmsda_exp_prev.so devis_amd/libmsda_hip.so
bash scripts/ab_bench.sh devis_amd/libmsda_exp_prev.so devis_amd/libmsda_hip.so -- --dtype bf16
python -m pytest tests/test_op_gpu.py tests/test_configs_gpu.py -m gpu -q 2>&1 | tail -3
