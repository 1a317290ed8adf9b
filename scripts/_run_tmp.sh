cd $GRAFT_REPO_ROOT
: > gpurun_out/r3k_ab.log
run() { MSDA_ENABLE_HOOKS=1 "$@" python3 bench.py --no-other-configs --no-cpu-baseline --steps 20 $EXTRA 2>>gpurun_out/r3k.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '$EXTRA', d['value'], d['ms_per_step'], {k[:18]:v['avg_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r3k_ab.log; }
EXTRA="--dtype bf16"
run env MSDA_GV_STORAGE=1 MSDA_SCATTER_DBG=0
run env MSDA_GV_STORAGE=0 MSDA_SCATTER_DBG=0
run env MSDA_GV_STORAGE=1 MSDA_SCATTER_DBG=4
run env MSDA_GV_STORAGE=0 MSDA_SCATTER_DBG=4
run env MSDA_GV_STORAGE=1 MSDA_SCATTER_DBG=8
run env MSDA_GV_STORAGE=0 MSDA_SCATTER_DBG=8
EXTRA=""
run env MSDA_SCATTER_DBG=4
cat gpurun_out/r3k_ab.log
