cd $GRAFT_REPO_ROOT
python -m pytest tests/test_op_gpu.py tests/test_configs_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r3o_tests.log
python3 scripts/check_cfg1.py > gpurun_out/r3o_cfg1.log 2>&1
MSDA_LIB=$PWD/devis_amd/libmsda_exp_nopipe.so python3 scripts/check_cfg1.py > gpurun_out/r3o_cfg1_nopipe.log 2>&1
cat gpurun_out/r3o_tests.log; grep -v "^ " gpurun_out/r3o_cfg1.log; echo NOPIPE; grep -v "^ " gpurun_out/r3o_cfg1_nopipe.log | head -4
