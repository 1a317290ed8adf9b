cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_window_gpu.py -m gpu -q -x 2>&1 | tail -2
for lib in libmsda_exp_prev libmsda_hip libmsda_exp_prev libmsda_hip; do
MSDA_LIB=$PWD/devis_amd/$lib.so python3 bench.py --no-other-configs --no-cpu-baseline --steps 5 --warmup 2 --clips 1 --queries 22223 --pyramid B --locs local --dtype f32 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', d['ms_per_step'], '  '.join('%s %.4f' % (k[:18], v['avg_ms']) for k,v in d['kernels'].items()))"
done
