"""GPU probe: resident-slab forward (MSDA_FWD_RS=1) against the previous kernels (MSDA_FWD_RS=0) on bench-shaped
inputs: max abs difference and HIP-event timings."""
import os, sys
os.environ.setdefault("MSDA_ENABLE_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from devis_amd import _native

def run(clips, pyr="A", locs="uniform", queries=300, dt="f32", layout="dense", reps=20):
    class A: pass
    a = A(); a.clips = clips; a.frames = 6; a.queries = queries; a.pyramid = pyr; a.locs = locs
    dtype = bench.DTYPES[dt]
    dev = torch.device("cuda:0")
    b = bench.make_clip_batch(a, dev, dtype, 1)
    T, q, M, D, L, P, W, S = b["dims"]
    if layout == "padded":
        buf = torch.zeros((b["value"].shape[0], S, M + 1, D), dtype=dtype, device=dev)
        buf[:, :, :M] = b["value"]; b["value"] = buf[:, :, :M]
    outs, times = {}, {}
    for mode in ("0", "1"):
        os.environ["MSDA_FWD_RS"] = mode
        _native.reload_knobs()
        out = torch.full((clips * T, q, M * D), float("nan"), dtype=dtype, device=dev)
        fn = lambda: _native.temporal_forward(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"],
                                              b["loc_t"], b["aw_t"], clips, out)
        for _ in range(3): fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        st = torch.cuda.current_stream()
        for s, e in ev:
            s.record(st); fn(); e.record(st)
        torch.cuda.synchronize()
        times[mode] = sum(s.elapsed_time(e) for s, e in ev) / reps
        outs[mode] = out.float().clone()
    os.environ.pop("MSDA_FWD_RS")
    _native.reload_knobs()
    diff = (outs["0"] - outs["1"]).abs().max().item()
    nan = int(torch.isnan(outs["1"]).sum())
    print("clips %3d pyr %s locs %-9s q %5d %s %-6s  old %.4f ms  rs %.4f ms  max|diff| %.3e  nan %d  scale %.3f"
          % (clips, pyr, locs, queries, dt, layout, times["0"], times["1"], diff, nan, outs["0"].abs().max().item()), flush=True)

if __name__ == "__main__":
    run(8, pyr="B"); run(8, pyr="B", dt="bf16"); run(16, pyr="B"); run(16, pyr="B", dt="bf16"); run(8, pyr="B", locs="clustered")
