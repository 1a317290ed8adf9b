cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_window_gpu.py tests/test_configs_gpu.py -m gpu -q 2>&1 | tail -2
MSDA_ENABLE_HOOKS=1 MSDA_WIN_MIN_HALO=8 timeout 900 python -m pytest tests/test_window_gpu.py -m gpu -q 2>&1 | tail -2
