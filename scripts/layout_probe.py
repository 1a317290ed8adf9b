"""GPU probe (round 5): forward / gather pass / scatter of the headline call with `value` dense (the reference's [N, S, M, D]),
row-padded (one spare head slot per pixel: what the modules' value_proj writes) and head-major ([M, N, S, D] memory)."""
import os
import sys

os.environ["MSDA_ENABLE_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from devis_amd import _native, tuning


def case(dt, layout, clips=16):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    sh = torch.tensor(tuning.PYRAMIDS["A"], dtype=torch.int64)
    L, S, T, W, M, D, P, q = 4, int(sh.prod(1).sum()), 6, 5, 8, 32, 4, 300
    G = clips * T
    rnd = lambda *s: torch.rand(*s, generator=g, device=dev)
    value = (rnd(G, S, M, D) * 2 - 1).to(dt)
    if layout == "head_major":
        value = _native.head_major(value)
    elif layout == "padded":
        buf = torch.zeros(G, S, M + 1, D, dtype=dt, device=dev)
        buf[:, :, :M] = value
        value = buf[:, :, :M]
    mk = lambda x: x.to(dt).contiguous()
    loc_c, loc_t = mk(rnd(G, q, M, L, P, 2)), mk(rnd(G, q, M, W * L, P, 2))
    aw = torch.softmax(torch.randn(G, q, M, L * P * (1 + W), generator=g, device=dev), -1)
    aw_c, aw_t = mk(aw[..., :L * P].reshape(G, q, M, L, P)), mk(aw[..., L * P:].reshape(G, q, M, W * L, P))
    go = mk(torch.randn(G, q, M * D, generator=g, device=dev))
    shapes, lsi = sh.to(dev), torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1])).to(dev)
    ftab = torch.tensor([[f for f in range(T) if f != t] for t in range(T)], dtype=torch.int32, device=dev)
    out = torch.empty(G, q, M * D, dtype=dt, device=dev)
    gv = torch.empty(G, S, M, D, dtype=torch.float32, device=dev)
    grads = [torch.empty_like(x) for x in (loc_c, aw_c, loc_t, aw_t)]
    ws = _native.bwd_workspace(dev, G, q, M, L * (1 + W))
    fwd = lambda: _native.temporal_forward(value, shapes, lsi, ftab, loc_c, aw_c, loc_t, aw_t, clips, out)
    bwd = lambda: _native.temporal_backward(value, shapes, lsi, ftab, loc_c, aw_c, loc_t, aw_t, go, clips, gv, *grads, ws)
    return fwd, bwd, ws


if __name__ == "__main__":
    for dt in (torch.float32, torch.bfloat16):
        for layout in ("dense", "padded", "head_major", "dense"):
            fwd, bwd, ws = case(dt, layout)
            t_f = tuning._time(fwd, 21)
            os.environ["MSDA_BWD_PHASES"] = "1"; _native.reload_knobs()
            t_g = tuning._time(bwd, 21)
            os.environ["MSDA_BWD_PHASES"] = "2"; _native.reload_knobs()
            t_s = tuning._time(lambda: (ws[:16].zero_(), bwd()), 21)
            os.environ.pop("MSDA_BWD_PHASES"); _native.reload_knobs()
            print("%-8s %-10s fwd %.4f gather %.4f scatter %.4f  sum %.4f ms" % (str(dt).split(".")[1], layout, t_f, t_g, t_s, t_f + t_g + t_s), flush=True)
            del fwd, bwd, ws
            torch.cuda.empty_cache()
