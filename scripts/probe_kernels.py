"""GPU probe (not part of the product): forward/backward kernel time vs clips and location patterns."""
import sys, torch, time
sys.argv = [sys.argv[0]] + sys.argv[1:]
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from devis_amd import _native

def run(clips, locs, pyramid="A", dtype="f32", reps=10):
    class A: pass
    a = A(); a.clips=clips; a.frames=6; a.queries=300; a.pyramid=pyramid; a.locs=locs
    dt = bench.DTYPES[dtype]
    dev = torch.device("cuda:0")
    b = bench.make_clip_batch(a, dev, dt, 1)
    if locs == "same":
        b["loc_c"].fill_(0.5); b["loc_t"].fill_(0.5)
    T,q,M,D,L,P,W,S = b["dims"]
    out = torch.empty((clips*T, q, M*D), dtype=dt, device=dev)
    gv = torch.zeros(b["value"].shape, dtype=torch.float32, device=dev)
    gl_c, ga_c = torch.empty_like(b["loc_c"]), torch.empty_like(b["aw_c"])
    gl_t, ga_t = torch.empty_like(b["loc_t"]), torch.empty_like(b["aw_t"])
    def tm(fn):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e)/reps*1e3
    f = tm(lambda: _native.temporal_forward(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"], clips, out))
    w = tm(lambda: _native.temporal_backward(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"], b["grad_out"], clips, gv, gl_c, ga_c, gl_t, ga_t)) if "--bwd" in sys.argv else float("nan")
    print(f"clips={clips:3d} locs={locs:9s} pyr={pyramid} {dtype}: fwd {f:9.1f} us ({f/clips:7.1f} us/clip)  bwd {w:10.1f} us ({w/clips:8.1f} us/clip)", flush=True)

def main():
    for dtype in ("f32", "bf16"):
        for locs in ("uniform", "clustered", "same"):
            for clips in (1, 4, 16, 32):
                run(clips, locs, dtype=dtype)
    run(8, "uniform", pyramid="B"); run(8, "clustered", pyramid="B")

if __name__ == "__main__":
    main()
