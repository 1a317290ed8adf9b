"""GPU: long randomised parity sweep (op + fused temporal op vs the CPU oracle) over shapes, dtypes, value
layouts and kernel routes -- a one-off soak beyond tests/test_fuzz_gpu.py.  Usage: fuzz_long.py [first] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import make_inputs, make_temporal_inputs, oracle_fwd_bwd, round_to, temporal_reference
from devis_amd import _native
from devis_amd.functions import MSDeformAttnFunction, MSDeformAttnTemporalFunction
DEV = "cuda:0"
os.environ["MSDA_ENABLE_HOOKS"] = "1"
from devis_amd import _native as _native_mod
ROUTES = [{}, {"MSDA_SCATTER_DBG": "16"}, {"MSDA_BWD_CULL": "2"}, {"MSDA_BWD_CULL": "0"}, {"MSDA_SCATTER_LDS_KB": "6", "MSDA_SCATTER_OWN": "0"},
          {"MSDA_SCATTER_OWN": "0"}, {"MSDA_FWD_RS": "1", "MSDA_BWD_RS": "1"}, {"MSDA_FWD_RS": "0", "MSDA_BWD_RS": "0"},
          {"MSDA_FWD_RS": "1", "MSDA_BWD_RS": "1", "MSDA_SCATTER_DBG": "16"}, {"MSDA_BWD_RS": "1", "MSDA_BWD_CULL": "0"},
          {"MSDA_BWD_MODE": "atomic"}, {"MSDA_FWD_RS": "1", "MSDA_FWD_RS_NT": "4"},
          {"MSDA_BWD_RS": "1", "MSDA_BWD_RS_FSPLIT": "1"}, {"MSDA_BWD_RS": "1", "MSDA_BWD_RS_FSPLIT": "2", "MSDA_FWD_RS": "1", "MSDA_FWD_RS_NT": "2"},
          {"MSDA_BWD_RS": "1", "MSDA_BWD_RS_FSPLIT": "4"}, {"MSDA_BWD_RS": "1", "MSDA_BWD_RS_TPW": "2", "MSDA_BWD_RS_FSPLIT": "0"},
          # the tile forward with 1 / 2 / 5 / 8 waves per tile (auto picks 3 or 1: the LDS clamp loop and the other counts ran nowhere)
          {"MSDA_FWD_RS": "0", "MSDA_FWD_TILE_WAVES": "1"}, {"MSDA_FWD_RS": "0", "MSDA_FWD_TILE_WAVES": "2"},
          {"MSDA_FWD_RS": "0", "MSDA_FWD_TILE_WAVES": "5"}, {"MSDA_FWD_RS": "0", "MSDA_FWD_TILE_WAVES": "8"},
          # round 6: the matrix-pipe scatter wherever it applies (D = 32, >= 2 levels, >= 16 queries, coarse levels <= 303 px), with and
          # without the resident-slab gather pass (which then leaves culling records for the owner kernel's levels only)
          {"MSDA_SCATTER_MFMA": "1"}, {"MSDA_SCATTER_MFMA": "1", "MSDA_BWD_RS": "1"}, {"MSDA_SCATTER_MFMA": "1", "MSDA_BWD_RS": "1", "MSDA_BWD_RS_FSPLIT": "2"},
          {"MSDA_SCATTER_MFMA": "1", "MSDA_BWD_RS": "0"}, {"MSDA_SCATTER_MFMA": "1", "MSDA_BWD_ALL_RECORDS": "1"}]
KEYS = ["MSDA_SCATTER_DBG", "MSDA_BWD_CULL", "MSDA_SCATTER_LDS_KB", "MSDA_SCATTER_OWN", "MSDA_FWD_RS", "MSDA_BWD_RS", "MSDA_BWD_MODE", "MSDA_FWD_RS_NT",
        "MSDA_BWD_RS_FSPLIT", "MSDA_BWD_RS_TPW", "MSDA_FWD_TILE_WAVES", "MSDA_SCATTER_MFMA", "MSDA_BWD_ALL_RECORDS"]
def maxabs(a, b): return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()) if a.size else 0.0
def layout(v, kind):
    if kind == 1: return _native.head_major(v)
    if kind == 2:
        buf = torch.full((v.shape[0], v.shape[1], v.shape[2] + 1, v.shape[3]), float("nan"), dtype=v.dtype, device=v.device)
        buf[:, :, :v.shape[2]] = v
        return buf[:, :, :v.shape[2]]
    return v
def shapes_of(rng, L, big):
    hi = (40, 60) if big else (14, 17)
    return [(int(rng.integers(1, hi[0])), int(rng.integers(1, hi[1]))) for _ in range(L)]
first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(50000 + seed)
    for k in KEYS: os.environ.pop(k, None)
    route = ROUTES[int(rng.integers(0, len(ROUTES)))]
    os.environ.update(route)
    _native_mod.reload_knobs()
    lay = int(rng.integers(0, 3))
    big = rng.random() < 0.25
    if seed % 7 == 3:
        # encoder-shaped: one query per pixel, forced onto the resident-window kernels
        from helpers import localise
        os.environ.update({"MSDA_FWD_WIN": "1", "MSDA_BWD_WIN": "1", "MSDA_WIN_MIN_HALO": str(int(rng.choice([3, 5, 9])))}); _native_mod.reload_knobs()
        L = int(rng.integers(1, 6)); h0, w0 = int(rng.integers(4, 48)), int(rng.integers(4, 48))
        shp = []
        for l in range(L):
            shp.append((h0, w0)); h0, w0 = max(1, (h0 + int(rng.integers(0, 2))) // 2), max(1, (w0 + int(rng.integers(0, 2))) // 2)
        S = sum(h * w for h, w in shp)
        T = int(rng.integers(1, 6)); W = int(rng.integers(1, 4)); Pc, Pt = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        ftab = rng.integers(0, T, size=(T, W)).astype(np.int32)
        tdt = [torch.float32, torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 4))]
        d = make_temporal_inputs(seed, T, W, 8, 32, S, shp, Pc, Pt, ftab=ftab, dtype=np.float32)
        sg = [None, 1.0, 2.5, 6.0][int(rng.integers(0, 4))]
        if sg is not None:
            d["loc_c"] = localise(d["loc_c"], shp, sg, seed + 1); d["loc_t"] = localise(d["loc_t"], shp, sg, seed + 2)
        if tdt != torch.float32: d = round_to(d, tdt)
        keys = ("value", "shapes", "lsi", "ftab", "loc_c", "aw_c", "loc_t", "aw_t", "grad_out")
        ref = temporal_reference(*(np.asarray(d[k], dtype=np.float64) if d[k].dtype.kind == "f" else d[k] for k in keys))
        f = lambda k: torch.from_numpy(np.asarray(d[k], dtype=np.float64)).to(DEV, tdt)
        v = layout(f("value"), lay).requires_grad_(True)
        lc, ac, lt, at = (f(k).requires_grad_(True) for k in ("loc_c", "aw_c", "loc_t", "aw_t"))
        out = MSDeformAttnTemporalFunction.apply(v, torch.from_numpy(d["shapes"]).to(DEV), torch.from_numpy(d["lsi"]).to(DEV),
                                                 torch.from_numpy(d["ftab"]).to(DEV), lc, ac, lt, at, 1)
        used = "resident-window" in _native.last_route()
        g = torch.autograd.grad(out, (v, lc, ac, lt, at), f("grad_out"))
        got = [t.detach().double().cpu().numpy() for t in (out,) + tuple(g)]
        errs = [maxabs(x, y) / max(1.0, np.abs(y).max()) for x, y in zip(got, ref)]
        cfg = dict(kind="window", shapes=shp, T=T, W=W, Pc=Pc, Pt=Pt, lay=lay, dtype=str(tdt), sigma=sg, used=used)
        ok = max(errs[0], errs[1], errs[3], errs[5]) <= {torch.float32: 1e-4, torch.bfloat16: 2e-2, torch.float16: 4e-3}[tdt]
        for k in ("MSDA_FWD_WIN", "MSDA_BWD_WIN", "MSDA_WIN_MIN_HALO"): os.environ.pop(k, None)
    elif seed % 2 == 0:
        D = int(rng.choice([4, 8, 16, 32, 32, 32, 64, 128, 12])); M = int(rng.choice([1, 2, 4, 8, 8, 16]))
        L, P = int(rng.integers(1, 6)), int(rng.integers(1, 7))
        N, Lq = int(rng.integers(1, 5)), int(rng.integers(1, 3000 if big else 80))
        d = make_inputs(seed, N, M, D, Lq, shapes_of(rng, L, big), P, "wide" if rng.random() < 0.7 else "unit", np.float32, value_scale=1.0)
        ref, ref64 = oracle_fwd_bwd(d, np.float32), oracle_fwd_bwd(d, np.float64)
        f = lambda k: torch.from_numpy(np.asarray(d[k], dtype=np.float64)).to(DEV, torch.float32)
        v = layout(f("value"), lay).requires_grad_(True); l = f("loc").requires_grad_(True); a = f("aw").requires_grad_(True)
        step = int(rng.choice([1, N, 64]))
        out = MSDeformAttnFunction.apply(v, torch.from_numpy(d["shapes"]).to(DEV), torch.from_numpy(d["lsi"]).to(DEV), l, a, step)
        g = torch.autograd.grad(out, (v, l, a), f("grad_out"))
        got = [t.detach().double().cpu().numpy() for t in (out,) + tuple(g)]
        errs = [maxabs(got[0], ref64[0]) / max(1.0, np.abs(ref64[0]).max())] + [maxabs(x, y) / max(1.0, np.abs(y).max()) for x, y in zip(got[1:], ref[1:])]
        cfg = dict(kind="op", D=D, M=M, L=L, P=P, N=N, Lq=Lq, big=big, lay=lay, route=route)
        ok = errs[0] <= 1e-5 and max(errs[1:]) <= 1e-4
    else:
        D = int(rng.choice([8, 16, 32, 32, 64])); M = int(rng.choice([2, 4, 8, 8]))
        L, Pc, Pt = int(rng.integers(1, 5)), int(rng.integers(1, 6)), int(rng.integers(1, 6))
        T = int(rng.integers(2, 7)); W = int(rng.integers(1, T))
        ftab = rng.integers(0, T, size=(T, W)).astype(np.int32)
        Lq = int(rng.integers(1, 900 if big else 60))
        clips = int(rng.integers(1, 4))
        shp = shapes_of(rng, L, big)
        tdt = [torch.float32, torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 4))]      # storage type
        ds = [make_temporal_inputs(seed * 7 + c, T, W, M, D, Lq, shp, Pc, Pt, ftab=ftab, dtype=np.float32) for c in range(clips)]
        loc32 = tdt != torch.float32 and rng.random() < 0.5       # ABI v11: float32 locations / weights beside a 16-bit value
        if tdt != torch.float32:
            ds = [dict(x, **round_to({k: v for k, v in x.items() if not loc32 or k in ("value", "grad_out")}, tdt)) for x in ds]
        keys = ("value", "shapes", "lsi", "ftab", "loc_c", "aw_c", "loc_t", "aw_t", "grad_out")
        refs = [temporal_reference(*(np.asarray(d[k], dtype=np.float64) if d[k].dtype.kind == "f" else d[k] for k in keys)) for d in ds]
        ref = [np.concatenate([r[i] for r in refs], 0) for i in range(6)]
        d = ds[0]
        f = lambda k: torch.from_numpy(np.concatenate([np.asarray(x[k], dtype=np.float64) for x in ds], 0)).to(DEV, tdt)
        v = layout(f("value"), lay).requires_grad_(True)
        fl = (lambda k: torch.from_numpy(np.concatenate([np.asarray(x[k], dtype=np.float32) for x in ds], 0)).to(DEV)) if loc32 else f
        lc, ac, lt, at = (fl(k).requires_grad_(True) for k in ("loc_c", "aw_c", "loc_t", "aw_t"))
        out = MSDeformAttnTemporalFunction.apply(v, torch.from_numpy(d["shapes"]).to(DEV), torch.from_numpy(d["lsi"]).to(DEV),
                                                 torch.from_numpy(d["ftab"]).to(DEV), lc, ac, lt, at, clips)
        g = torch.autograd.grad(out, (v, lc, ac, lt, at), f("grad_out"))
        got = [t.detach().double().cpu().numpy() for t in (out,) + tuple(g)]
        errs = [maxabs(x, y) / max(1.0, np.abs(y).max()) for x, y in zip(got, ref)]
        cfg = dict(kind="temporal", D=D, M=M, L=L, Pc=Pc, Pt=Pt, T=T, W=W, Lq=Lq, clips=clips, big=big, lay=lay, route=route, dtype=str(tdt), loc32=loc32)
        # grad_loc (indices 2, 4) vs an fp64 reference flips cells at pixel borders: judged loosely
        ok = max(errs[0], errs[1]) <= {torch.float32: 1e-4, torch.bfloat16: 2e-2, torch.float16: 4e-3}[tdt] and \
            max(errs[3], errs[5]) <= (1e-4 if loc32 else {torch.float32: 1e-4, torch.bfloat16: 2e-2, torch.float16: 4e-3}[tdt])
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, cfg, ["%.2e" % e for e in errs], flush=True)
torch.cuda.synchronize()
print("fuzz_long: seeds %d..%d done, %d mismatches" % (first, first + count - 1, bad))
