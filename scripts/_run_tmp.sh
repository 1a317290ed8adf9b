cd $GRAFT_REPO_ROOT
python -m pytest tests/test_op_gpu.py tests/test_configs_gpu.py -m gpu -q -x -k "alternate or overwritten or bench_scale or round2 or golden or storage" 2>&1 | tail -3
bash scripts/ab_bench.sh devis_amd/libmsda_exp_ch1.so devis_amd/libmsda_hip.so devis_amd/libmsda_exp_ch4.so
bash scripts/ab_bench.sh devis_amd/libmsda_exp_ch1.so devis_amd/libmsda_hip.so -- --dtype bf16 | head -2
