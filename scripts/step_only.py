"""GPU probe: 3 fused forward + backward launches (for rocprofv3 --pmc passes)."""
import os, sys
os.environ.setdefault("MSDA_ENABLE_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from devis_amd import _native
class A: pass
a = A(); a.clips=int(os.environ.get("CLIPS","16")); a.frames=6; a.queries=int(os.environ.get("QUERIES","300")); a.sampling="storage"; a.pyramid=os.environ.get("PYR","A"); a.locs=os.environ.get("LOCS","uniform")
dt = bench.DTYPES[os.environ.get("DT","f32")]
dev = torch.device("cuda:0")
b = bench.make_clip_batch(a, dev, dt, 1)
T,q,M,D,L,P,W,S = b["dims"]
out = torch.empty((a.clips*T, q, M*D), dtype=dt, device=dev)
gv = torch.zeros(b["value"].shape, dtype=torch.float32, device=dev)
gl_c, ga_c = torch.empty_like(b["loc_c"]), torch.empty_like(b["aw_c"])
gl_t, ga_t = torch.empty_like(b["loc_t"]), torch.empty_like(b["aw_t"])
for _ in range(3):
    _native.temporal_forward(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"], a.clips, out)
    _native.temporal_backward(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"], b["grad_out"], a.clips, gv, gl_c, ga_c, gl_t, ga_t)
torch.cuda.synchronize()
print("done")
