cd $GRAFT_REPO_ROOT
bash scripts/ab_bench.sh devis_amd/libmsda_exp_prev.so devis_amd/libmsda_hip.so
python -m pytest tests/test_op_gpu.py tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -2
