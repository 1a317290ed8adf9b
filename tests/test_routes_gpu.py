"""Which kernel family a call takes (msda_last_route) for the shapes the route rules of round 4 were audited on
(scripts/route_audit.py): a change of a rule that moves one of these shapes to another family shows up
here, not only as a slower bench line.  Values do not matter (zeros): routes depend on sizes only."""
import os

import numpy as np
import pytest
import torch

from helpers import PYR_A, make_temporal_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def rules_only(request):
    """The route expectations below are those of the RULES (csrc/msda_api.hip): the shipped table of measured routes
    (devis_amd/routes.json) is taken out for them and put back afterwards; the tests of the table itself keep it."""
    from devis_amd import _native
    _native.load()
    if request.node.name.startswith("test_shipped_routes"):
        yield
        return
    _native.clear_routes()
    try:
        yield
    finally:
        _native.clear_routes()
        _native._load_shipped_routes()

PYR = {"A": [(45, 80), (23, 40), (12, 20), (6, 10)], "S": [(60, 96), (30, 48), (15, 24), (8, 12)], "B": [(100, 167), (50, 84), (25, 42), (13, 21)]}
DEV = "cuda:0"


def _temporal(pyr, clips, Lq, dtype, T=6):
    from devis_amd import _native
    shapes = torch.tensor(PYR[pyr], dtype=torch.int64, device=DEV)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S, M, D, L, P, W = int(shapes.prod(1).sum()), 8, 32, 4, 4, T - 1
    if Lq is None:
        Lq = S
    ftab = torch.tensor([[t for t in range(T) if t != f] for f in range(T)], dtype=torch.int32, device=DEV)
    value = torch.zeros(clips * T, S, M, D, dtype=dtype, device=DEV)
    loc_c = torch.full((clips * T, Lq, M, L, P, 2), 0.5, dtype=dtype, device=DEV)
    aw_c = torch.zeros(clips * T, Lq, M, L, P, dtype=dtype, device=DEV)
    loc_t = torch.full((clips * T, Lq, M, W * L, P, 2), 0.5, dtype=dtype, device=DEV)
    aw_t = torch.zeros(clips * T, Lq, M, W * L, P, dtype=dtype, device=DEV)
    out = torch.empty(clips * T, Lq, M * D, dtype=dtype, device=DEV)
    _native.temporal_forward(value, shapes, lsi, ftab, loc_c, aw_c, loc_t, aw_t, clips, out)
    fwd = _native.last_route()
    gv = torch.empty(value.shape, dtype=_native.grad_value_dtype(value, shapes, Lq, L, P, clips=clips, window=W, Pt=P), device=DEV)
    grads = [torch.empty_like(x) for x in (loc_c, aw_c, loc_t, aw_t)]
    _native.temporal_backward(value, shapes, lsi, ftab, loc_c, aw_c, loc_t, aw_t, torch.zeros_like(out), clips, gv, *grads)
    torch.cuda.synchronize()
    return fwd, _native.last_route()


@pytest.mark.parametrize("pyr,clips,Lq,dtype,fwd_has,bwd_has", [
    # the call DeVIS issues: tile forward, gather pass on the slab kernel with the frames as a workgroup index
    ("A", 1, 300, torch.float32, "(tile kernel)", "one source frame per workgroup"),
    ("B", 1, 300, torch.bfloat16, "tile kernel", "one source frame per workgroup"),
    # ... at the query counts of DeVIS's shipped configs (60 per frame on YouTube-VIS, 180 on OVIS): 2.5-3x faster than the tile kernels
    ("A", 1, 60, torch.float32, "tile kernel, several waves per tile", "one source frame per workgroup"),
    ("S", 1, 180, torch.float16, "tile kernel, several waves per tile", "one source frame per workgroup"),
    # the bench batch
    ("A", 16, 300, torch.float32, "resident-slab kernel, 1 tiles per wave", "one source frame per workgroup"),
    ("A", 16, 300, torch.bfloat16, "resident-slab kernel, 4 tiles per wave", "one source frame per workgroup"),
    # large maps outside the slab: no frame split (12-20 % slower there)
    ("B", 16, 300, torch.float32, "resident-slab kernel, 1 tiles per wave", "resident-slab kernel, grad_loc/grad_attn)"),
    ("S", 16, 300, torch.float32, "resident-slab kernel, 1 tiles per wave", "resident-slab kernel, grad_loc/grad_attn)"),
    ("S", 16, 300, torch.bfloat16, "resident-slab kernel", "one source frame per workgroup"),
    # encoder-shaped calls: slab kernels while three levels fit, window kernels when a 4-byte slab holds two or fewer
    ("A", 1, None, torch.float32, "resident-slab kernel, 2 tiles per wave", "resident-slab kernel, grad_loc/grad_attn)"),
    ("A", 1, None, torch.bfloat16, "resident-slab kernel, 4 tiles per wave", "resident-slab kernel, grad_loc/grad_attn)"),
    ("S", 1, None, torch.float32, "resident-window kernel", "resident-window kernel"),
    ("S", 1, None, torch.bfloat16, "resident-slab kernel, 2 tiles per wave", "resident-slab kernel, grad_loc/grad_attn)"),
    # four clips in fp32: one round of (clip, head, part) workgroups -> the frame-split grid; not in 2-byte types
    ("A", 4, 300, torch.float32, "resident-slab kernel, 1 tiles per wave", "one source frame per workgroup"),
    ("A", 4, 300, torch.bfloat16, "resident-slab kernel, 1 tiles per wave", "resident-slab kernel, grad_loc/grad_attn)"),
], ids=lambda v: str(v).replace("torch.", "") if not isinstance(v, str) or len(v) < 3 else None)
def test_temporal_call_routes(pyr, clips, Lq, dtype, fwd_has, bwd_has):
    fwd, bwd = _temporal(pyr, clips, Lq, dtype)
    assert fwd_has in fwd, fwd
    assert bwd_has in bwd, bwd
    assert "owner-computes scatter" in bwd, bwd
    # the zero-fill rides in the scatter kernel when every level fits a band
    assert "zero-fill" not in bwd, bwd


def test_single_frame_decoder_like_call_on_a_sparse_fp32_slab_takes_the_tile_forward():
    """36 images x 300 queries on the SwinL pyramid in fp32: the slab would start at level 2 and 19 tiles would sit on 2 x 16
    waves -- the tile forward is 30 % faster there (the gather pass stays on the slab kernel)."""
    from devis_amd import _native
    shapes = torch.tensor(PYR["S"], dtype=torch.int64, device=DEV)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S, N, Lq = int(shapes.prod(1).sum()), 36, 300
    value = torch.zeros(N, S, 8, 32, device=DEV)
    loc = torch.full((N, Lq, 8, 4, 4, 2), 0.5, device=DEV)
    aw = torch.zeros(N, Lq, 8, 4, 4, device=DEV)
    out = torch.empty(N, Lq, 256, device=DEV)
    _native.forward(value, shapes, lsi, loc, aw, out)
    assert "tile kernel" in _native.last_route(), _native.last_route()
    value_a = torch.zeros(N, 4820, 8, 32, device=DEV)
    shapes_a = torch.tensor(PYR["A"], dtype=torch.int64, device=DEV)
    lsi_a = torch.cat((shapes_a.new_zeros(1), shapes_a.prod(1).cumsum(0)[:-1]))
    _native.forward(value_a, shapes_a, lsi_a, loc, aw, out)
    assert "resident-slab kernel" in _native.last_route(), _native.last_route()


def test_pinned_route_is_taken_and_changes_nothing_but_the_kernel():
    """msda_pin_route (ABI v12): a pinned setting reroutes calls of exactly that shape -- msda_last_route shows it --, results
    are the rule-chosen route's, a knob forced through the environment still wins, removing the pin restores the rules."""
    from devis_amd import _native
    d = make_temporal_inputs(21, 6, 5, 8, 32, 300, PYR_A, 4, 4, dtype=np.float32)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in d.items() if isinstance(v, np.ndarray)}
    v = t["value"].float().contiguous()
    shapes, lsi, ftab = t["shapes"], t["lsi"], t["ftab"].int()
    lc, ac, lt, at = (t[k].float().contiguous() for k in ("loc_c", "aw_c", "loc_t", "aw_t"))
    out_a, out_b = (torch.empty(6, 300, 256, device=DEV) for _ in range(2))
    key = _native.route_key(False, 0, 1, 6, 5, v.shape[1], 8, 32, 4, 300, 4, 4, shapes.cpu())
    try:
        _native.pin_route(key, "")
        _native.temporal_forward(v, shapes, lsi, ftab, lc, ac, lt, at, 1, out_a)
        auto = _native.last_route()
        other = {"fwd_rs": 1, "fwd_rs_nt": 1, "fwd_win": 0} if "tile kernel" in auto else {"fwd_rs": 0, "fwd_win": 0}
        _native.pin_route(key, other)
        _native.temporal_forward(v, shapes, lsi, ftab, lc, ac, lt, at, 1, out_b)
        pinned = _native.last_route()
        assert pinned != auto and ("tile kernel" in pinned) != ("tile kernel" in auto), (auto, pinned)
        assert torch.allclose(out_a, out_b, rtol=1e-5, atol=1e-6)
        # another shape (two clips) is not affected
        v2 = torch.cat([v, v]); cat2 = lambda x: torch.cat([x, x])
        out2 = torch.empty(12, 300, 256, device=DEV)
        _native.temporal_forward(v2, shapes, lsi, ftab, cat2(lc), cat2(ac), cat2(lt), cat2(at), 2, out2)
        assert torch.allclose(out2[:6], out_a, rtol=1e-5, atol=1e-6)
    finally:
        _native.pin_route(key, "")
    _native.temporal_forward(v, shapes, lsi, ftab, lc, ac, lt, at, 1, out_b)
    assert _native.last_route() == auto


@pytest.mark.parametrize("case", list(range(8)))
def test_shipped_routes_are_within_five_percent_of_the_best_alternative(case):
    """for the audited shapes the route the library takes on its own (rules + devis_amd/routes.json) is as fast as
    the fastest route it could be forced onto -- measured here, on this box, on a sample of the audit's shapes (the full list
    runs in `python -m devis_amd.tuning --audit`).  The table pins every alternative the audit found 5 % ahead of the rules; this
    test fails from 8 % on (two audits of the same build on two boxes disagreed on a third of the 3-5 % pins: that band is noise),
    after a second, longer look.  Differences under 3 us are not held against a route."""
    import devis_amd.tuning as tuning
    from devis_amd import _native
    if not os.path.exists(_native.ROUTES_FILE):
        pytest.skip("no shipped route table")
    shapes = [s for s in tuning.audit_shapes(quick=True)]
    picks = shapes[case::8][:2]
    def behind(r):
        out = []
        for side, auto_key, times_key in (("forward", "auto_ms", "times"), ("backward", "gather_auto_ms", "gather_times"),
                                          ("backward", "scatter_auto_ms", "scatter_times")):
            auto, times = r[side][auto_key], r[side][times_key]
            if times and auto > 1.08 * min(times.values()) + 0.003:
                out.append((auto_key, auto, times))
        return out

    for pyr, kind, clips, q, dt, l32 in picks:
        args = dict(clips=clips, Lq=q, kind=kind, pin=False, keep_pins=True, sampling_fp32=l32)
        bad = behind(tuning.tune(tuning.PYRAMIDS[pyr], dt, reps=11, **args))
        if bad:         # a clock ramp or a neighbour on the box can cost one measurement 5 %: what must hold is the second look
            bad = [b for b in behind(tuning.tune(tuning.PYRAMIDS[pyr], dt, reps=25, **args)) if b[0] in {x[0] for x in bad}]
        assert not bad, (pyr, kind, clips, q, dt, bad)
