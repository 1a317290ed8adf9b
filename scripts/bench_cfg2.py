"""GPU: BASELINE.json configs[1] -- single-frame Deformable-DETR encoder attention on the 800x1333 pyramid:
Lq = S = 22223, M=8, K=4, C=256, bf16 (also f32), batch N images.  Sampling locations are what an
encoder produces: the query's own pixel centre plus offsets of a few pixels (N(0, (2 px)^2) per level).
Prints one JSON line with per-kernel times and algorithmic-byte roofline fractions.  Not the headline bench."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devis_amd import _native

PYR_B = [(100, 167), (50, 84), (25, 42), (13, 21)]
N = int(os.environ.get("N", "8"))
dev = torch.device("cuda:0")
res = {}
for name, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
    g = torch.Generator().manual_seed(0)
    shapes = torch.tensor(PYR_B)
    S = int(shapes.prod(1).sum()); M, D, L, P = 8, 32, 4, 4
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    # reference point of query s = centre of its own pixel, in normalised coordinates
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij"), -1)
                     .reshape(-1, 2).flip(-1) for h, w in PYR_B], 0)                       # [S, 2] (x, y)
    wh = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
    loc = ref[None, :, None, None, None, :] + torch.randn(N, S, M, L, P, 2, generator=g) * 2.0 / wh[None, None, None, :, None, :]
    value = torch.rand(N, S, M, D, generator=g) * 2 - 1
    aw = torch.softmax(torch.randn(N, S, M, L * P, generator=g), -1).view(N, S, M, L, P)
    go = torch.randn(N, S, M * D, generator=g)
    value, loc, aw, go = (x.to(dev, dt).contiguous() for x in (value, loc, aw, go))
    shapes, lsi = shapes.to(dev), lsi.to(dev)
    out = torch.empty(N, S, M * D, dtype=dt, device=dev)
    gv = torch.zeros(value.shape, dtype=torch.float32, device=dev)
    gl, ga = torch.empty_like(loc), torch.empty_like(aw)

    def tm(fn, reps=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / reps

    fwd = tm(lambda: _native.forward(value, shapes, lsi, loc, aw, out))
    os.environ["MSDA_BWD_PHASES"] = "1"
    gat = tm(lambda: _native.backward(value, shapes, lsi, loc, aw, go, gv, gl, ga))
    os.environ["MSDA_BWD_PHASES"] = "2"
    sca = tm(lambda: _native.backward(value, shapes, lsi, loc, aw, go, gv, gl, ga))
    os.environ.pop("MSDA_BWD_PHASES")
    e = value.element_size(); C = M * D; pts = N * S * M * L * P
    b_fwd = N * S * C * e + pts * 3 * e + N * S * C * e          # value once + loc/attn + out
    b_gat = N * S * C * e + N * S * C * e + pts * 3 * e * 2
    b_sca = pts * 3 * e + N * S * C * e + N * S * C * 4
    res[name] = {"fwd_ms": round(fwd, 4), "gather_ms": round(gat, 4), "scatter_ms": round(sca, 4),
                 "fwd_GBps": round(b_fwd / fwd / 1e6, 1), "fwd_frac_of_8TBps": round(b_fwd / fwd / 1e6 / 8000, 4),
                 "gather_frac": round(b_gat / gat / 1e6 / 8000, 4), "scatter_frac": round(b_sca / sca / 1e6 / 8000, 4),
                 "fwd_M_queries_per_s": round(N * S / fwd / 1e3, 2),
                 "fwd_bwd_M_queries_per_s": round(N * S / (fwd + gat + sca) / 1e3, 2)}
print(json.dumps({"workload": "cfg2 encoder single-frame MSDeformAttn, 800x1333 pyramid, Lq=S=22223, M=8, K=4, C=256, N=%d images, "
                              "locations = own pixel centre + N(0,(2px)^2)" % N, "results": res}))
