"""GPU: the resident-window kernels (devis_amd/csrc/msda_win.hip; encoder-shaped calls: one query per pixel, local sampling)
against the CPU oracle -- forced through the test knobs and on the automatic route, with sampling that stays inside the
windows, that leaves them often (second pass from memory) and that ignores locality altogether."""
import numpy as np
import pytest
import torch

from helpers import (PYR_A, localise, make_inputs, make_temporal_inputs, oracle_fwd_bwd, round_to, temporal_reference)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PYR_S = [(20, 33), (10, 17), (5, 9), (3, 5)]            # S = 890: 3 x 5 tiles of 8 x 8, ragged edges, odd level ratios
_maxabs = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()) if a.size else 0.0


def _force(monkeypatch, fwd=True, bwd=True):
    monkeypatch.setenv("MSDA_FWD_WIN", "1" if fwd else "0")
    monkeypatch.setenv("MSDA_BWD_WIN", "1" if bwd else "0")


def _route():
    from devis_amd import _native
    return _native.last_route()


@pytest.mark.parametrize("sigma", [1.5, 6.0, None], ids=["local", "wide-tails", "uniform"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 1e-2), (torch.float16, 1.5e-3)], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("shapes,T", [(PYR_S, 3), (PYR_A, 2)], ids=["small", "pyrA"])
def test_temporal_encoder_call_on_the_window_kernels(shapes, T, dtype, tol, sigma, monkeypatch):
    from devis_amd.functions import MSDeformAttnTemporalFunction
    _force(monkeypatch)
    S = int(sum(h * w for h, w in shapes))
    d = make_temporal_inputs(7, T=T, W=T - 1, M=8, D=32, Lq=S, shapes=shapes, Pc=4, Pt=4)
    if sigma is not None:
        d["loc_c"] = localise(d["loc_c"], shapes, sigma, 1)
        d["loc_t"] = localise(d["loc_t"], shapes, sigma, 2)
    d = round_to({k: (np.asarray(v, dtype=np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}, dtype)
    keys = ("value", "shapes", "lsi", "ftab", "loc_c", "aw_c", "loc_t", "aw_t", "grad_out")
    ref = temporal_reference(*(d[k] for k in keys))
    # (grad_loc of fp32 runs: against the oracle in fp32 arithmetic -- a location that lands on a pixel border in fp32 but not in
    # fp64 selects another cell, tests/test_op_gpu.py)
    ref32 = temporal_reference(*(np.asarray(d[k], dtype=np.float32) if d[k].dtype.kind == "f" else d[k] for k in keys)) \
        if dtype == torch.float32 else None
    f = lambda k: torch.from_numpy(d[k]).to(DEV, dtype).requires_grad_(True)
    leaves = [f(k) for k in ("value", "loc_c", "aw_c", "loc_t", "aw_t")]
    out = MSDeformAttnTemporalFunction.apply(leaves[0], torch.from_numpy(d["shapes"]).to(DEV), torch.from_numpy(d["lsi"]).to(DEV),
                                             torch.from_numpy(d["ftab"]).to(DEV), *leaves[1:], 1)
    assert "resident-window" in _route(), _route()
    scale = lambda x: max(1.0, float(np.abs(x).max()))
    err = _maxabs(out.detach().double().cpu().numpy(), ref[0])
    assert err <= tol * scale(ref[0]), err
    grads = torch.autograd.grad(out, leaves, torch.from_numpy(d["grad_out"]).to(DEV, dtype))
    for i, (g, r) in enumerate(zip(grads, ref[1:])):
        bound = ((2e-4 if i in (1, 3) else 2e-5) if dtype == torch.float32 else 3 * tol) * scale(r)
        if i in (1, 3) and dtype == torch.float32:
            r = ref32[1 + i]
        err = _maxabs(g.double().cpu().numpy(), r)
        assert err <= bound, (i, err, bound)


@pytest.mark.parametrize("P,L", [(4, 4), (3, 3), (1, 2), (6, 4)], ids=["P4L4", "P3L3", "P1L2", "P6L4"])
@pytest.mark.parametrize("sigma", [1.5, None], ids=["local", "wide"])
def test_plain_encoder_call_on_the_window_kernels(P, L, sigma, monkeypatch):
    """The single-frame encoder's MSDeformAttn (N images, Lq = S): other point / level counts take the narrow point loads
    and groups that straddle levels."""
    from devis_amd.functions import MSDeformAttnFunction
    _force(monkeypatch)
    shapes = PYR_S[:L]
    S = int(sum(h * w for h, w in shapes))
    d = make_inputs(11, 2, 8, 32, S, shapes, P, "wide", np.float32, value_scale=1.0)
    if sigma is not None:
        d["loc"] = localise(d["loc"], shapes, sigma, 3)
    ref = oracle_fwd_bwd(d, np.float64)
    f = lambda k: torch.from_numpy(d[k]).to(DEV).requires_grad_(True)
    v, loc, aw = f("value"), f("loc"), f("aw")
    out = MSDeformAttnFunction.apply(v, torch.from_numpy(d["shapes"]).to(DEV), torch.from_numpy(d["lsi"]).to(DEV), loc, aw, 64)
    assert "resident-window" in _route(), _route()
    scale = lambda x: max(1.0, float(np.abs(x).max()))
    assert _maxabs(out.detach().double().cpu().numpy(), ref[0]) <= 2e-6 * scale(ref[0])
    gv, gl, ga = torch.autograd.grad(out, (v, loc, aw), torch.from_numpy(d["grad_out"]).to(DEV))
    ref32 = oracle_fwd_bwd(d, np.float32)
    assert _maxabs(gv.cpu().numpy(), ref[1]) <= 2e-5 * scale(ref[1])
    assert _maxabs(gl.cpu().numpy(), ref32[2]) <= 2e-4 * scale(ref32[2])
    assert _maxabs(ga.cpu().numpy(), ref[3]) <= 2e-5 * scale(ref[3])


@pytest.mark.parametrize("layout", ["padded", "head-major"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 1e-2)], ids=["f32", "bf16"])
def test_window_kernels_take_the_other_value_layouts(layout, dtype, tol, monkeypatch):
    """`value` with one spare head slot per pixel row (what the modules' value_proj writes) and head-major `value`: the
    window staging and the second pass address pixels through the strides."""
    from devis_amd import _native
    from devis_amd.functions import MSDeformAttnTemporalFunction
    _force(monkeypatch)
    shapes, T = PYR_S, 3
    S = int(sum(h * w for h, w in shapes))
    d = make_temporal_inputs(21, T=T, W=T - 1, M=8, D=32, Lq=S, shapes=shapes, Pc=4, Pt=4)
    d["loc_c"] = localise(d["loc_c"], shapes, 4.0, 1)
    d["loc_t"] = localise(d["loc_t"], shapes, 4.0, 2)
    d = round_to({k: (np.asarray(v, dtype=np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}, dtype)
    keys = ("value", "shapes", "lsi", "ftab", "loc_c", "aw_c", "loc_t", "aw_t", "grad_out")
    ref = temporal_reference(*(d[k] for k in keys))
    f = lambda k: torch.from_numpy(d[k]).to(DEV, dtype)
    v = f("value")
    if layout == "padded":
        buf = torch.full((T, S, 9, 32), float("nan"), dtype=dtype, device=DEV)
        buf[:, :, :8] = v
        v = buf[:, :, :8]
    else:
        v = _native.head_major(v)
    v = v.requires_grad_(True)
    leaves = [v] + [f(k).requires_grad_(True) for k in ("loc_c", "aw_c", "loc_t", "aw_t")]
    out = MSDeformAttnTemporalFunction.apply(v, torch.from_numpy(d["shapes"]).to(DEV), torch.from_numpy(d["lsi"]).to(DEV),
                                             torch.from_numpy(d["ftab"]).to(DEV), *leaves[1:], 1)
    assert "resident-window" in _route(), _route()
    scale = lambda x: max(1.0, float(np.abs(x).max()))
    assert _maxabs(out.detach().double().cpu().numpy(), ref[0]) <= tol * scale(ref[0])
    grads = torch.autograd.grad(out, leaves, f("grad_out"))
    for i in (0, 2, 4):         # grad_value, grad_attn (current / temporal)
        assert _maxabs(grads[i].double().cpu().numpy(), ref[1 + i]) <= (2e-5 if dtype == torch.float32 else 3 * tol) * scale(ref[1 + i]), i


def test_window_route_is_automatic_where_the_slab_holds_the_last_level_only(route_rules_only):
    """Automatic route: one query per pixel AND a pyramid of which the resident-slab kernels could keep the last level at most
    (fp32 at 800x1333).  (The gather pass follows the same rule; the backward runs on an autograd thread, whose route string
    this thread cannot read -- bench.py names the kernels of both directions.)"""
    from devis_amd.functions import MSDeformAttnFunction
    from helpers import PYR_B
    SB = int(sum(h * w for h, w in PYR_B)); SS = int(sum(h * w for h, w in PYR_S))
    for shapes, Lq, dtype, expect in ((PYR_B, SB, torch.float32, True), (PYR_B, SB, torch.bfloat16, False), (PYR_B, 300, torch.float32, False),
                                      (PYR_S, SS, torch.float32, False)):
        d = make_inputs(5, 1, 8, 32, Lq, shapes, 4, "unit", np.float32)
        f = lambda k: torch.from_numpy(d[k]).to(DEV, dtype if d[k].dtype.kind == "f" else None)
        v, loc, aw = f("value").requires_grad_(True), f("loc").requires_grad_(True), f("aw").requires_grad_(True)
        out = MSDeformAttnFunction.apply(v, f("shapes"), f("lsi"), loc, aw, 64)
        assert ("forward (resident-window" in _route()) == expect, (Lq, dtype, _route())


@pytest.mark.parametrize("seed", range(24))
def test_window_kernels_random_pyramids(seed, monkeypatch):
    """Random pyramids (ragged level ratios, levels smaller than a tile, 1-5 levels), frame tables with repeated / missing
    frames, point counts, dtypes and sampling spreads, forced onto the window kernels."""
    if seed % 3 == 0:
        monkeypatch.setenv("MSDA_WIN_MIN_HALO", "9")        # plans with two staging phases (narrow halos otherwise fit in one)
    from devis_amd.functions import MSDeformAttnTemporalFunction
    _force(monkeypatch)
    rng = np.random.default_rng(900 + seed)
    L = int(rng.integers(1, 6))
    h0, w0 = int(rng.integers(6, 40)), int(rng.integers(6, 40))
    shapes = []
    for l in range(L):
        shapes.append((h0, w0))
        h0, w0 = max(1, (h0 + int(rng.integers(0, 2))) // 2), max(1, (w0 + int(rng.integers(0, 2))) // 2)
    S = int(sum(h * w for h, w in shapes))
    T = int(rng.integers(1, 5)); W = int(rng.integers(1, 4))
    ftab = rng.integers(0, T, size=(T, W)).astype(np.int32)
    Pc, Pt = int(rng.integers(1, 6)), int(rng.integers(1, 6))
    dtype = [torch.float32, torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 4))]
    loc32 = dtype != torch.float32 and rng.random() < 0.5
    d = make_temporal_inputs(seed, T, W, 8, 32, S, shapes, Pc, Pt, ftab=ftab)
    sigma = [None, 1.0, 3.0, 8.0][int(rng.integers(0, 4))]
    if sigma is not None:
        d["loc_c"] = localise(d["loc_c"], shapes, sigma, seed + 1)
        d["loc_t"] = localise(d["loc_t"], shapes, sigma, seed + 2)
    keep = ("value", "grad_out") if loc32 else tuple(d)
    d.update(round_to({k: np.asarray(d[k], dtype=np.float64) for k in keep if d[k].dtype.kind == "f"}, dtype))
    keys = ("value", "shapes", "lsi", "ftab", "loc_c", "aw_c", "loc_t", "aw_t", "grad_out")
    ref = temporal_reference(*(np.asarray(d[k], dtype=np.float64) if d[k].dtype.kind == "f" else d[k] for k in keys))
    ldt = torch.float32 if loc32 else dtype
    mk = lambda k, t: torch.from_numpy(np.asarray(d[k], dtype=np.float64)).to(DEV, t).requires_grad_(True)
    leaves = [mk("value", dtype)] + [mk(k, ldt) for k in ("loc_c", "aw_c", "loc_t", "aw_t")]
    out = MSDeformAttnTemporalFunction.apply(leaves[0], torch.from_numpy(d["shapes"]).to(DEV), torch.from_numpy(d["lsi"]).to(DEV),
                                             torch.from_numpy(d["ftab"]).to(DEV), *leaves[1:], 1)
    assert "resident-window" in _route(), _route()
    grads = torch.autograd.grad(out, leaves, torch.from_numpy(np.asarray(d["grad_out"], dtype=np.float64)).to(DEV, dtype))
    tol = {torch.float32: 1e-4, torch.bfloat16: 2e-2, torch.float16: 4e-3}[dtype]
    scale = lambda x: max(1.0, float(np.abs(x).max()))
    got = [out.detach()] + list(grads)
    for i, (g, r) in enumerate(zip(got, ref)):
        if i in (2, 4) and (dtype == torch.float32 or loc32):
            continue            # fp32 grad_loc vs an fp64 reference: cell flips at pixel borders (checked in the tests above)
        bound = (1e-4 if (loc32 and i in (3, 5)) else tol) * scale(r)
        err = _maxabs(g.double().cpu().numpy(), r)
        assert err <= bound, (i, err, bound, shapes, T, W, Pc, Pt, dtype, sigma)
