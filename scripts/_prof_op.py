import os, sys, cProfile, pstats, io, time
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import torch, bench
from devis_amd.functions import MSDeformAttnTemporalFunction
class A: pass
a = A(); a.clips, a.frames, a.queries, a.pyramid, a.locs, a.sampling = 1, 6, 300, "A", "uniform", "storage"
b = bench.make_clip_batch(a, torch.device("cuda:0"), torch.float32, 1)
leaves = [b[k].requires_grad_(True) for k in ("value", "loc_c", "aw_c", "loc_t", "aw_t")]
def fwd():
    with torch.no_grad():
        return MSDeformAttnTemporalFunction.apply(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"], 1)
def step():
    out = MSDeformAttnTemporalFunction.apply(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"], 1)
    torch.autograd.grad(out, leaves, b["grad_out"])
for _ in range(50): step()
torch.cuda.synchronize()
for name, fn, n in (("fwd", fwd, 3000), ("step", step, 2000)):
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(name, "host us per call (enqueue only): %.1f, with sync %.1f" % ((t1 - t0) / n * 1e6, (time.perf_counter() - t0) / n * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): fwd()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:4000])
