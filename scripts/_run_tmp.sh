cd $GRAFT_REPO_ROOT
python -m pytest tests/test_op_gpu.py -m gpu -q -k "graph_replay or storage_type" 2>&1 | tail -3 > gpurun_out/r3l_tests.log
: > gpurun_out/r3l_ab.log
run() { MSDA_ENABLE_HOOKS=1 "$@" python3 bench.py --no-other-configs --no-cpu-baseline --steps 20 $EXTRA 2>>gpurun_out/r3l.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '$EXTRA', d['value'], d['ms_per_step'], {k[:18]:v['avg_ms'] for k,v in d['kernels'].items()}, d.get('padded_value_layout'))" >> gpurun_out/r3l_ab.log; }
EXTRA=""
for lv in 1 2 3 4; do run env MSDA_SCATTER_DBG=$((32*lv)); done
for lv in 1 2; do run env MSDA_SCATTER_DBG=$((32*lv+1)); done
for lv in 1 2; do run env MSDA_SCATTER_DBG=$((32*lv+8)); done
cat gpurun_out/r3l_ab.log; cat gpurun_out/r3l_tests.log; tail -3 gpurun_out/r3l.err
python3 bench.py --steps 20 --no-cpu-baseline > gpurun_out/r3l_bench.json 2>gpurun_out/r3l_bench.err; python3 -c "
import json
d=json.loads(open('gpurun_out/r3l_bench.json').read().strip().splitlines()[-1]); o=d['other_configs']
print(d['value'], d['padded_value_layout']); print({k:o[k] for k in ('single_clip_latency','single_clip_graph')}); print(o['cfg1_encoder_800x1333_bf16']['kernels'])"
