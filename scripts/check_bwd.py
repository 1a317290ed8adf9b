"""GPU probe: gather-pass variants (MSDA_BWD_RS) -- timing and agreement of grad_loc / grad_attn / grad_value."""
import os, sys
os.environ.setdefault("MSDA_ENABLE_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from devis_amd import _native

def run(clips, env, pyr="A", locs="uniform", queries=300, reps=10, layout="dense"):
    class A: pass
    a = A(); a.clips = clips; a.frames = 6; a.queries = queries; a.pyramid = pyr; a.locs = locs
    dev = torch.device("cuda:0")
    b = bench.make_clip_batch(a, dev, torch.float32, 1)
    T, q, M, D, L, P, W, S = b["dims"]
    if layout == "padded":
        buf = torch.zeros((b["value"].shape[0], S, M + 1, D), dtype=torch.float32, device=dev)
        buf[:, :, :M] = b["value"]; b["value"] = buf[:, :, :M]
    gv = torch.zeros((clips * T, S, M, D), dtype=torch.float32, device=dev)
    outs = [torch.full_like(b[k], float("nan")) for k in ("loc_c", "aw_c", "loc_t", "aw_t")]
    ws = _native.bwd_workspace(dev, clips * T, q, M, L * (1 + W))
    def bwd():
        _native.temporal_backward(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"],
                                  b["grad_out"], clips, gv, *outs, workspace=ws)
    os.environ.update(env); os.environ["MSDA_BWD_PHASES"] = "3"; _native.reload_knobs()
    bwd(); torch.cuda.synchronize()
    route = _native.last_route()
    res = [gv.clone()] + [o.clone() for o in outs]
    os.environ["MSDA_BWD_PHASES"] = "1"; _native.reload_knobs()
    for _ in range(3): bwd()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    st = torch.cuda.current_stream()
    for s, e in ev:
        s.record(st); bwd(); e.record(st)
    torch.cuda.synchronize()
    ms = sum(s.elapsed_time(e) for s, e in ev) / reps
    for k in env: os.environ.pop(k)
    os.environ.pop("MSDA_BWD_PHASES"); _native.reload_knobs()
    return ms, res, route

if __name__ == "__main__":
    for clips, locs, pyr, q, layout in ((16, "uniform", "A", 300, "dense"), (16, "uniform", "A", 300, "padded"), (16, "clustered", "A", 300, "dense"),
                                        (8, "uniform", "A", 300, "dense"), (32, "uniform", "A", 300, "dense"), (2, "local", "A", 4820, "dense")):
        base = None
        for name, env in (("slab/tile", {"MSDA_BWD_RS": "0"}), ("resident-slab", {"MSDA_BWD_RS": "1"})):
            ms, res, route = run(clips, env, locs=locs, pyr=pyr, queries=q, layout=layout)
            if base is None: base = res
            diffs = ["%.1e" % ((r - b0).abs().max().item() / max(1.0, b0.abs().max().item())) for r, b0 in zip(res, base)]
            print("clips %2d %-9s pyr %s q %5d %-6s %-14s gather pass %.4f ms  rel diffs (gv, gl_c, ga_c, gl_t, ga_t) %s  nan %d" %
                  (clips, locs, pyr, q, layout, name, ms, " ".join(diffs), sum(int(torch.isnan(r).sum()) for r in res)), flush=True)
        print("   route:", route)
