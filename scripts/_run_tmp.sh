cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_window_gpu.py -m gpu -q 2>&1 | grep -E "^FAILED|^E   +Assert|passed|failed" | cut -c 1-300 | tail -30
