import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from devis_amd import _native
class A: pass
for mode in ("normal", "own-frame-only"):
    a = A(); a.clips=16; a.frames=6; a.queries=300; a.pyramid="A"; a.locs="uniform"
    dev = torch.device("cuda:0"); b = bench.make_clip_batch(a, dev, torch.float32, 1)
    if mode != "normal":
        b["ftab"] = torch.arange(6, dtype=torch.int32, device=dev)[:, None].repeat(1, 5).contiguous()
    T,q,M,D,L,P,W,S = b["dims"]
    out = torch.empty((a.clips*T, q, M*D), device=dev)
    def tm(fn, reps=10):
        for _ in range(2): fn()
        torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/reps*1e3
    f = tm(lambda: _native.temporal_forward(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"], a.clips, out))
    print(f"{mode}: fwd {f/16:.1f} us/clip", flush=True)
