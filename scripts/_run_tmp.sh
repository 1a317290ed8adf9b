cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -4
python3 bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d['other_configs']
print(d['value']); print({k:{a:b for a,b in o[k].items() if a!='workload'} for k in ('single_clip_latency','single_clip_graph','headline_bf16','cfg4_swinl_fp16_decoder_like','cfg4_mask_head_like_fp16')})"
