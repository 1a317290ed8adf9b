cd $GRAFT_REPO_ROOT
for sm in storage fp32; do
python3 bench.py --no-other-configs --no-cpu-baseline --steps 30 --dtype bf16 --sampling $sm 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$sm', d['value'], d['ms_per_step'], {k:v for k,v in d.get('kernels_ms',{}).items()} if 'kernels_ms' in d else [ (k, d[k]) for k in d if 'kernel' in k][:3])"
done
python -m pytest tests/test_configs_gpu.py -m gpu -q -k "reduced_precision" 2>&1 | tail -2
