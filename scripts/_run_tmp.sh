cd $GRAFT_REPO_ROOT
python3 bench.py --no-cpu-baseline --steps 20 2>&1 | tail -1 > gpurun_out/bench_win.json
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_win.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(json.dumps(d['other_configs']['temporal_encoder_800x1333_f32'])[:900])
print(json.dumps(d['other_configs']['temporal_encoder_layer'])[:400])
print(json.dumps(d['other_configs']['cfg1_encoder_800x1333_bf16'])[:900])
PY
python -m pytest tests -m gpu -q 2>&1 | tail -3
