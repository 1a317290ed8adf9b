"""GPU probe: fused forward / gather-pass time when only level 0 (45x80) exists (24 taps/row instead of 96)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from devis_amd import _native
bench.PYRAMIDS["Z"] = [(45, 80)]
bench.PYRAMIDS["Y"] = [(23, 40), (12, 20), (6, 10)]
class A: pass
for pyr in ("A", "Z", "Y"):
    a = A(); a.clips=16; a.frames=6; a.queries=300; a.pyramid=pyr; a.locs="uniform"
    dev = torch.device("cuda:0"); b = bench.make_clip_batch(a, dev, torch.float32, 1)
    # make_clip_batch hardcodes L from the pyramid
    T,q,M,D,L,P,W,S = b["dims"]
    out = torch.empty((a.clips*T, q, M*D), device=dev)
    gv = torch.zeros(b["value"].shape, device=dev)
    gl_c, ga_c = torch.empty_like(b["loc_c"]), torch.empty_like(b["aw_c"]); gl_t, ga_t = torch.empty_like(b["loc_t"]), torch.empty_like(b["aw_t"])
    def tm(fn, reps=10):
        for _ in range(2): fn()
        torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/reps*1e3
    f = tm(lambda: _native.temporal_forward(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"], a.clips, out))
    os.environ["MSDA_BWD_PHASES"]="1"
    g = tm(lambda: _native.temporal_backward(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"], b["grad_out"], a.clips, gv, gl_c, ga_c, gl_t, ga_t))
    os.environ["MSDA_BWD_PHASES"]="2"
    sc = tm(lambda: _native.temporal_backward(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"], b["grad_out"], a.clips, gv, gl_c, ga_c, gl_t, ga_t))
    os.environ.pop("MSDA_BWD_PHASES")
    print(f"pyramid {pyr} L={L} S={S}: fwd {f/16:.1f} us/clip  gather {g/16:.1f}  scatter {sc/16:.1f}", flush=True)
