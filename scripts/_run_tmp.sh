cd $GRAFT_REPO_ROOT
python -m pytest tests/test_op_gpu.py tests/test_configs_gpu.py tests/test_modules_gpu.py -m gpu -q -x 2>&1 | tail -3
run() { MSDA_ENABLE_HOOKS=1 "$@" python3 bench.py --no-other-configs --no-cpu-baseline --steps 30 $EXTRA 2>>gpurun_out/r3q.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '$EXTRA', d['value'], d['ms_per_step'], {k[:18]:v['avg_ms'] for k,v in d['kernels'].items()})"; }
EXTRA="--clips 1 --queries 4820 --locs local"
for w in 0 256 512 768 1024 1536; do run env MSDA_SCATTER_DBG=$w; done
EXTRA="--clips 2 --queries 4820 --locs local"
for w in 0 256 512; do run env MSDA_SCATTER_DBG=$w; done
EXTRA="--clips 1 --queries 4820 --locs uniform"
for w in 0 256; do run env MSDA_SCATTER_DBG=$w; done
