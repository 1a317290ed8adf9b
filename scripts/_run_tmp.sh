cd $GRAFT_REPO_ROOT
python -m pytest tests/test_op_gpu.py tests/test_configs_gpu.py -m gpu -q -x --deselect tests/test_configs_gpu.py::test_config_sized_modules_reduced_precision_vs_reference_fixture 2>&1 | tail -5 > gpurun_out/r3f_tests.log
: > gpurun_out/r3f_ab.log
run() { MSDA_ENABLE_HOOKS=1 "$@" python3 bench.py --no-other-configs --no-cpu-baseline --steps 30 $EXTRA 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '$EXTRA', d['value'], {k[:18]:v['avg_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r3f_ab.log; }
for rep in 1 2; do
EXTRA=""
run env MSDA_FWD_RS_NT=1 MSDA_BWD_RS_TPW=1
run env MSDA_FWD_RS_NT=1 MSDA_BWD_RS_TPW=2
run env MSDA_FWD_RS_NT=1 MSDA_BWD_RS_TPW=4
EXTRA="--dtype bf16"
run env MSDA_FWD_RS_NT=2 MSDA_BWD_RS_TPW=1
run env MSDA_FWD_RS_NT=2 MSDA_BWD_RS_TPW=2
run env MSDA_FWD_RS_NT=4 MSDA_BWD_RS_TPW=4
done
EXTRA="--value-layout padded"
run env MSDA_FWD_RS_NT=1 MSDA_BWD_RS_TPW=1
run env MSDA_FWD_RS_NT=1 MSDA_BWD_RS_TPW=2
EXTRA="--clips 64"
run env MSDA_FWD_RS_NT=1 MSDA_BWD_RS_TPW=1
run env MSDA_FWD_RS_NT=2 MSDA_BWD_RS_TPW=2
EXTRA="--clips 4"
run env MSDA_FWD_RS_NT=1 MSDA_BWD_RS_TPW=1
cat gpurun_out/r3f_ab.log; cat gpurun_out/r3f_tests.log
