"""GPU probe: scatter-pass timing for kernel variants (MSDA_SCATTER_DBG bits 8..9 select the owner-computes variant;
MSDA_SCATTER_OWN=0 is the LDS-atomic scatter)."""
import os, sys
os.environ.setdefault("MSDA_ENABLE_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from devis_amd import _native

def run(clips, env, pyr="A", locs="uniform", queries=300, reps=10):
    class A: pass
    a = A(); a.clips = clips; a.frames = 6; a.queries = queries; a.pyramid = pyr; a.locs = locs
    dev = torch.device("cuda:0")
    b = bench.make_clip_batch(a, dev, torch.float32, 1)
    T, q, M, D, L, P, W, S = b["dims"]
    gv = torch.zeros(b["value"].shape, dtype=torch.float32, device=dev)
    gl_c, ga_c = torch.empty_like(b["loc_c"]), torch.empty_like(b["aw_c"])
    gl_t, ga_t = torch.empty_like(b["loc_t"]), torch.empty_like(b["aw_t"])
    ws = _native.bwd_workspace(dev, clips * T, q, M, L * (1 + W))
    def bwd():
        ws[:16].zero_()
        _native.temporal_backward(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"],
                                  b["grad_out"], clips, gv, gl_c, ga_c, gl_t, ga_t, workspace=ws)
    os.environ.update(env); os.environ["MSDA_BWD_PHASES"] = "3"; _native.reload_knobs()
    bwd(); torch.cuda.synchronize()
    ref = gv.clone()
    os.environ["MSDA_BWD_PHASES"] = "2"; _native.reload_knobs()
    for _ in range(3): bwd()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    st = torch.cuda.current_stream()
    for s, e in ev:
        s.record(st); bwd(); e.record(st)
    torch.cuda.synchronize()
    ms = sum(s.elapsed_time(e) for s, e in ev) / reps
    for k in env: os.environ.pop(k)
    os.environ.pop("MSDA_BWD_PHASES"); _native.reload_knobs()
    return ms, ref

if __name__ == "__main__":
    for clips, locs, pyr, q in ((16, "uniform", "A", 300), (1, "local", "A", 4820), (1, "local", "B", 22223), (1, "uniform", "B", 22223)):
        base = None
        for name, env in (("lds-atomic", {"MSDA_SCATTER_OWN": "0"}), ("own", {"MSDA_SCATTER_DBG": "0"})):
            ms, gv = run(clips, env, locs=locs, pyr=pyr, queries=q)
            if base is None: base = gv
            print("clips %2d %-9s pyr %s q %5d %-18s scatter %.4f ms   max|diff vs lds-atomic| %.3e (scale %.3f)" % (clips, locs, pyr, q, name, ms, (gv - base).abs().max().item(), base.abs().max().item()), flush=True)
