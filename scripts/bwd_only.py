"""GPU probe: a few fused backward launches only (for rocprofv3 --pmc passes)."""
import os, sys
os.environ.setdefault("MSDA_ENABLE_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from devis_amd import _native
class A: pass
a = A(); a.clips=int(os.environ.get("CLIPS","16")); a.frames=6; a.queries=300; a.pyramid=os.environ.get("PYR","A"); a.locs=os.environ.get("LOCS","uniform")
dt = bench.DTYPES[os.environ.get("DT","f32")]
dev = torch.device("cuda:0")
b = bench.make_clip_batch(a, dev, dt, 1)
gv = torch.zeros(b["value"].shape, dtype=torch.float32, device=dev)
gl_c, ga_c = torch.empty_like(b["loc_c"]), torch.empty_like(b["aw_c"])
gl_t, ga_t = torch.empty_like(b["loc_t"]), torch.empty_like(b["aw_t"])
T, M, W, L = a.frames, b["value"].shape[2], b["ftab"].shape[1], b["shapes"].shape[0]
for _ in range(3):
    ws = _native.bwd_workspace(dev, a.clips * T, a.queries, M, L * (1 + W))
    _native.temporal_backward(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"], b["grad_out"], a.clips, gv, gl_c, ga_c, gl_t, ga_t, workspace=ws)
torch.cuda.synchronize()
print("done")
