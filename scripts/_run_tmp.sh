cd $GRAFT_REPO_ROOT
python3 bench.py --no-other-configs --no-cpu-baseline --steps 30 --dtype bf16 > gpurun_out/r3i_bf16.json 2> gpurun_out/r3i_bf16.err; tail -5 gpurun_out/r3i_bf16.err; tail -c 1500 gpurun_out/r3i_bf16.json
python -m pytest tests/test_configs_gpu.py tests/test_op_gpu.py -m gpu -q -k "storage_type or bench_scale or reduced_precision_vs_reference" 2>&1 | tail -5
