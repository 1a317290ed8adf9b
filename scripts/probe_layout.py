"""GPU probe: kernel times with the dense vs the head-major value layout (include/msda.h value_strides)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from devis_amd import _native
class A: pass
a = A(); a.clips=int(os.environ.get("CLIPS","16")); a.frames=6; a.queries=int(os.environ.get("Q","300")); a.pyramid=os.environ.get("PYR","A"); a.locs=os.environ.get("LOCS","uniform")
dev = torch.device("cuda:0")
b = bench.make_clip_batch(a, dev, torch.float32, 1)
T, M, W, L = a.frames, b["value"].shape[2], b["ftab"].shape[1], b["shapes"].shape[0]
out = torch.empty(b["value"].shape[0], a.queries, b["value"].shape[2] * b["value"].shape[3], device=dev)
gv = torch.empty(b["value"].shape, device=dev)
gl_c, ga_c = torch.empty_like(b["loc_c"]), torch.empty_like(b["aw_c"])
gl_t, ga_t = torch.empty_like(b["loc_t"]), torch.empty_like(b["aw_t"])
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
def padded(v, extra_heads=1):
    """dense, but every pixel row padded by `extra_heads` head slots: pixel stride (M + extra) * D"""
    G, S, M, D = v.shape
    buf = torch.zeros(G, S, M + extra_heads, D, dtype=v.dtype, device=v.device)
    buf[:, :, :M] = v
    return buf[:, :, :M]
for name, v in (("dense", b["value"]), ("head-major", _native.head_major(b["value"])), ("padded +1 head", padded(b["value"])),
                ("padded +3 heads", padded(b["value"], 3))):
    fwd = lambda: _native.temporal_forward(v, b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"], a.clips, out)
    def bwd():
        ws = _native.bwd_workspace(dev, a.clips * T, a.queries, M, L * (1 + W))
        _native.temporal_backward(v, b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"], b["grad_out"], a.clips, gv, gl_c, ga_c, gl_t, ga_t, workspace=ws)
    res = {"fwd": timeit(fwd)}
    os.environ["MSDA_BWD_PHASES"] = "1"; res["gather"] = timeit(bwd)
    os.environ["MSDA_BWD_PHASES"] = "2"; res["scatter"] = timeit(bwd)
    os.environ.pop("MSDA_BWD_PHASES")
    print(name, {k: round(x, 4) for k, x in res.items()})
print("transpose copy ms", round(timeit(lambda: _native.head_major(b["value"])), 4))
