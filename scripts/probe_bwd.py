"""GPU probe: backward variants (MSDA_BWD_MODE / LDS budget) timing on the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from devis_amd import _native
class A: pass
def run(clips, locs, pyramid="A", dtype="f32", reps=5, env=None):
    for k, v in (env or {}).items(): os.environ[k] = v
    a = A(); a.clips=clips; a.frames=6; a.queries=300; a.pyramid=pyramid; a.locs=locs
    dt = bench.DTYPES[dtype]; dev = torch.device("cuda:0")
    b = bench.make_clip_batch(a, dev, dt, 1)
    gv = torch.zeros(b["value"].shape, dtype=torch.float32, device=dev)
    gl_c, ga_c = torch.empty_like(b["loc_c"]), torch.empty_like(b["aw_c"])
    gl_t, ga_t = torch.empty_like(b["loc_t"]), torch.empty_like(b["aw_t"])
    fn = lambda: _native.temporal_backward(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"], b["grad_out"], clips, gv, gl_c, ga_c, gl_t, ga_t)
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    w = s.elapsed_time(e)/reps*1e3
    print(f"clips={clips:3d} locs={locs:9s} pyr={pyramid} {dtype} env={env}: bwd {w:10.1f} us ({w/clips:8.1f} us/clip)", flush=True)
    for k in (env or {}): os.environ.pop(k)
if __name__ == "__main__":
    for env in ({"MSDA_SCATTER_LDS_KB":"64","MSDA_SCATTER_WG_PER_CU":"2"}, {"MSDA_SCATTER_LDS_KB":"32","MSDA_SCATTER_WG_PER_CU":"4"},
                {"MSDA_SCATTER_LDS_KB":"128","MSDA_SCATTER_WG_PER_CU":"1"}, {"MSDA_SCATTER_LDS_KB":"64","MSDA_SCATTER_WG_PER_CU":"4"}):
        for clips in (1, 16):
            run(clips, "uniform", env=env)
    run(16, "clustered"); run(16, "uniform", dtype="bf16"); run(8, "uniform", pyramid="B")
    run(4, "uniform", env={"MSDA_BWD_MODE":"atomic"}, reps=2)
