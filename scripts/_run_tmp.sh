cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_window_gpu.py tests/test_op_gpu.py -m gpu -q 2>&1 | tail -2
python3 bench.py --no-cpu-baseline --steps 20 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['other_configs']['single_clip_latency'], d['other_configs']['temporal_encoder_800x1333_f32']['fwd_bwd_ms'])"
