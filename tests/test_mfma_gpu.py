"""GPU parity of the matrix-pipe scatter (round 6, devis_amd/csrc/msda_mfma.hip): grad_value of the coarse pyramid levels as a
split-precision bf16 / f16 matrix product, FORCED (MSDA_SCATTER_MFMA=1) on shapes that hit every kernel size, against the CPU
oracle (the reference's atomicAdd scatter, ms_deform_im2col_cuda.cuh:87-159, restated in oracle/).  Everything goes through the
C ABI; msda_last_route() must name the kernel, so a silent fall-back to the owner-computes scatter fails the test."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import PYR_A, PYR_B, make_inputs, make_temporal_inputs, oracle_fwd_bwd, round_to, temporal_reference
from test_op_gpu import DEV, _maxabs, _run_op, _run_temporal

pytestmark = pytest.mark.gpu

KEYS = ("value", "shapes", "lsi", "ftab", "loc_c", "aw_c", "loc_t", "aw_t", "grad_out")
SWIN = [(60, 96), (30, 48), (15, 24), (8, 12)]


def _ref(d, dt=np.float64):
    return temporal_reference(*(np.asarray(d[k], dtype=dt) if d[k].dtype.kind == "f" else d[k] for k in KEYS))


def _routes_of_backward(fn):
    """Run `fn` (a forward + backward through autograd) and return what msda_last_route() said on the backward's thread."""
    from devis_amd import _native
    seen = []
    orig = _native.temporal_backward, _native.backward

    def tb(*a, **k):
        orig[0](*a, **k)
        seen.append(_native.last_route())

    def pb(*a, **k):
        orig[1](*a, **k)
        seen.append(_native.last_route())
    _native.temporal_backward, _native.backward = tb, pb
    try:
        out = fn()
    finally:
        _native.temporal_backward, _native.backward = orig
    return out, seen


# (pyramid, T, window, Lq, Pc, Pt): coarse part -> kernel size.  Two fused levels (12x20 + 6x10 = 300 px: 10 tiles), one level of
# 273 / 96 / 60 / 20 px (10 / 4 / 4 / 2 tiles), a level too large for any kernel behind one that fits, Lq = 16 (one step), 17 (the
# overlapping last step), a multiple of 16, fewer points in one of the sources, windows with repeated and with missing frames
SHAPES = [
    ("A", PYR_A, 3, 2, 37, 4, 4),
    ("A-lq16", PYR_A, 2, 1, 16, 4, 4),
    ("A-lq17", PYR_A, 2, 1, 17, 4, 3),
    ("A-lq64", PYR_A, 2, 1, 64, 2, 4),
    ("B", PYR_B, 2, 1, 33, 4, 4),
    ("swin", SWIN, 2, 1, 29, 4, 4),
    ("two-small", [(9, 11), (6, 10), (4, 5)], 4, 3, 41, 4, 2),           # 60 + 20 px: 4 tiles, two levels
    ("one-small", [(30, 30), (20, 19), (5, 4)], 3, 2, 50, 3, 4),         # 380 px does not fit with the 20: only the last level
    ("three", [(7, 9), (5, 6), (3, 4)], 5, 4, 23, 4, 4),
]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 8e-3), (torch.float16, 1e-3)], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("name,pyr,T,W,Lq,Pc,Pt", SHAPES, ids=[s[0] for s in SHAPES])
def test_forced_matrix_pipe_scatter_against_the_oracle(name, pyr, T, W, Lq, Pc, Pt, dtype, tol, monkeypatch):
    monkeypatch.setenv("MSDA_SCATTER_MFMA", "1")
    rng = np.random.default_rng(hash(name) % 1000)
    ftab = None if W == T - 1 else rng.integers(0, T, size=(T, W)).astype(np.int32)
    d = make_temporal_inputs(500 + len(name), T, W, 8, 32, Lq, pyr, Pc, Pt, ftab=ftab, dtype=np.float64)
    if dtype != torch.float32:
        d = round_to(d, dtype)
    else:
        d = {k: (v.astype(np.float32) if v.dtype.kind == "f" else v) for k, v in d.items()}
    got, routes = _routes_of_backward(lambda: _run_temporal(d, dtype))
    assert routes and "matrix-pipe" in routes[0] and "owner-computes" in routes[0], routes
    ref = _ref(d)
    gv, want = got[1], ref[1]
    scale = max(1e-30, np.abs(want).max())
    assert _maxabs(gv, want) <= tol * scale, (name, _maxabs(gv, want), scale)
    # the coarse levels themselves (the part the new kernel wrote), against their own scale
    shapes = np.asarray(pyr)
    first = int((shapes[:-1, 0] * shapes[:-1, 1]).sum())
    sub, wsub = gv[:, first:], want[:, first:]
    assert _maxabs(sub, wsub) <= tol * max(1e-30, np.abs(wsub).max()), name
    for i in (3, 5):                                                      # grad_attn is untouched by the route
        assert _maxabs(got[i], ref[i]) <= max(tol, 1e-4) * max(1.0, np.abs(ref[i]).max())


def test_matrix_pipe_scatter_batch_of_clips_and_route_independence(monkeypatch):
    """16 clips of the bench shape in one call, forced on and forced off: same grad_value to rounding, clip 0 and 15 against the
    oracle; and one clip alone (48 items) forced on."""
    T, Lq = 6, 300
    ds = [make_temporal_inputs(700 + c, T, 5, 8, 32, Lq, PYR_A, 4, 4) for c in range(16)]
    cat = {k: (np.concatenate([x[k] for x in ds], 0) if k not in ("shapes", "lsi", "ftab") else ds[0][k]) for k in ds[0]}
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MSDA_SCATTER_MFMA", mode)
        res[mode], routes = _routes_of_backward(lambda: _run_temporal(cat, torch.float32, clips=16))
        assert ("matrix-pipe" in routes[0]) == (mode == "1"), routes
    scale = np.abs(res["0"][1]).max()
    assert _maxabs(res["1"][1], res["0"][1]) <= 2e-5 * scale
    for c in (0, 15):
        ref = _ref(ds[c])
        assert _maxabs(res["1"][1][c * T:(c + 1) * T], ref[1]) <= 2e-5 * max(1e-30, np.abs(ref[1]).max())
    monkeypatch.setenv("MSDA_SCATTER_MFMA", "1")
    one, routes = _routes_of_backward(lambda: _run_temporal(ds[3], torch.float32))
    assert "matrix-pipe" in routes[0]
    ref = _ref(ds[3])
    assert _maxabs(one[1], ref[1]) <= 2e-5 * max(1e-30, np.abs(ref[1]).max())


def test_matrix_pipe_scatter_is_automatic_on_the_bench_batch_and_off_for_one_clip(monkeypatch):
    monkeypatch.delenv("MSDA_SCATTER_MFMA", raising=False)
    from devis_amd import _native
    _native.clear_routes()
    try:
        ds = [make_temporal_inputs(800 + c, 6, 5, 8, 32, 300, PYR_A, 4, 4) for c in range(4)]
        cat = {k: (np.concatenate([x[k] for x in ds], 0) if k not in ("shapes", "lsi", "ftab") else ds[0][k]) for k in ds[0]}
        _, routes = _routes_of_backward(lambda: _run_temporal(cat, torch.float32, clips=4))
        assert "matrix-pipe" in routes[0], routes                         # 4 clips x 6 frames x 8 heads = 192 items
        _, routes = _routes_of_backward(lambda: _run_temporal(ds[0], torch.float32))
        assert "matrix-pipe" not in routes[0], routes                     # 48 items: the owner-computes kernel keeps every level
    finally:
        _native._load_shipped_routes()


def test_gather_pass_leaves_records_only_for_the_owner_kernels_levels(monkeypatch):
    """Round 6: with the coarse levels on the matrix pipe the gather pass writes culling records for levels [0, l0) only
    (msda_api.hip: the plan is made before the gather pass).  The same results with every record written
    (MSDA_BWD_ALL_RECORDS=1, a measurement hook), on a decoder batch that takes the route by itself."""
    from devis_amd import _native
    monkeypatch.delenv("MSDA_SCATTER_MFMA", raising=False)
    _native.clear_routes()
    try:
        ds = [make_temporal_inputs(900 + c, 6, 5, 8, 32, 300, PYR_A, 4, 4) for c in range(4)]
        cat = {k: (np.concatenate([x[k] for x in ds], 0) if k not in ("shapes", "lsi", "ftab") else ds[0][k]) for k in ds[0]}
        got, routes = _routes_of_backward(lambda: _run_temporal(cat, torch.float32, clips=4))
        assert "matrix-pipe" in routes[0], routes
        monkeypatch.setenv("MSDA_BWD_ALL_RECORDS", "1")
        _native.reload_knobs()
        every = _run_temporal(cat, torch.float32, clips=4)
        for i, (a, b) in enumerate(zip(got, every)):
            if i == 1:      # grad_value: the owner-computes kernel's lists are linked in arrival order -- last-bit differences run to run
                assert _maxabs(a, b) <= 1e-6 * np.abs(b).max()
                lsi2 = int(cat["lsi"][2])
                assert np.array_equal(a[:, lsi2:], b[:, lsi2:])            # the matrix-pipe levels: a fixed summation order
            else:
                assert np.array_equal(a, b)
        ref = _ref(ds[2])
        n = ds[2]["value"].shape[0]
        assert _maxabs(got[1][2 * n:3 * n], ref[1]) <= 2e-5 * np.abs(ref[1]).max()
    finally:
        monkeypatch.delenv("MSDA_BWD_ALL_RECORDS", raising=False)
        _native.reload_knobs()
        _native._load_shipped_routes()


def test_matrix_pipe_scatter_plain_op_duplicates_and_borders(monkeypatch):
    """The plain operator (no frame table, one source), with what the merge must get right: all points of a group on ONE pixel
    cell, points exactly on cell borders and on the map's edge rows / columns, points outside the map, zero attention."""
    monkeypatch.setenv("MSDA_SCATTER_MFMA", "1")
    pyr = [(16, 20), (12, 20), (6, 10)]
    d = make_inputs(11, 3, 8, 32, 48, pyr, 4, "wide", np.float32, value_scale=1.0)
    loc = d["loc"]
    loc[:, 0:8, :, :, :, :] = loc[:, 0:8, :, :, 0:1, :]                                   # four identical points
    loc[:, 8:16, :, :, 1:, :] = loc[:, 8:16, :, :, 0:1, :] + np.float32(0.02)             # overlapping footprints
    for li, (h, w) in enumerate(pyr):                                                     # exact cell borders: x * W - 0.5 integer
        loc[:, 16:20, :, li, :, 0] = (np.arange(4, dtype=np.float32)[None, None, :] + 0.5) / w
        loc[:, 16:20, :, li, :, 1] = (np.arange(4, dtype=np.float32)[None, None, :] + 1.5) / h
        loc[:, 20:22, :, li, :, 0] = np.float32(0.25) / w                                 # w_im = -0.25: left column only
        loc[:, 22:24, :, li, :, 1] = np.float32(1.0) - np.float32(0.25) / h               # h_im = H - 0.75: last row only
    loc[:, 24:28] = np.float32(-0.5)                                                      # outside: skipped (cuh:288)
    d["aw"][:, 28:32] = 0.0
    got, routes = _routes_of_backward(lambda: _run_op(d, torch.float32))
    assert "matrix-pipe" in routes[0], routes
    ref = oracle_fwd_bwd(d, np.float32)
    ref64 = oracle_fwd_bwd(d, np.float64)
    assert _maxabs(got[1], ref64[1]) <= 2e-5 * np.abs(ref64[1]).max()
    for a, b in zip(got[2:], ref[2:]):
        assert _maxabs(a, b) <= 2e-5 * max(1.0, np.abs(b).max())


def test_matrix_pipe_scatter_fills_nan_when_the_host_copy_of_the_shapes_lied():
    """include/msda.h: for backward calls spatial_shapes_host MUST be a true copy.  A copy that shows coarse levels small enough
    for the matrix-pipe kernel while the DEVICE shapes are larger does not return silently wrong sums: the kernel sees the
    device shapes and fills those levels with NaN."""
    from devis_amd import _native
    import os
    lib = _native.load()
    real = [(10, 12), (19, 20)]                                            # 380 px: no matrix-pipe kernel holds the last level
    d = make_inputs(5, 2, 8, 32, 40, real, 4, "unit", np.float32, value_scale=1.0)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in d.items()}
    N, S, M, D = t["value"].shape
    _, Lq, _, L, P, _ = t["loc"].shape
    gv = torch.zeros(t["value"].shape, device=DEV)
    gl, ga = torch.empty_like(t["loc"]), torch.empty_like(t["aw"])
    ws = _native.bwd_workspace(DEV, N, Lq, M, L)
    lie = (ctypes.c_int64 * 4)(10, 12, 6, 10)
    os.environ["MSDA_SCATTER_MFMA"] = "1"
    _native.reload_knobs()
    try:
        rc = lib.msda_backward(0, t["value"].data_ptr(), t["shapes"].data_ptr(), t["lsi"].data_ptr(), t["loc"].data_ptr(),
                               t["aw"].data_ptr(), t["grad_out"].data_ptr(), N, S, M, D, L, Lq, P, gv.data_ptr(), 0, gl.data_ptr(),
                               ga.data_ptr(), ws.data_ptr(), ws.numel() * 4, None, lie, torch.cuda.current_stream().cuda_stream)
        route = _native.last_route()
    finally:
        os.environ.pop("MSDA_SCATTER_MFMA", None)
        _native.reload_knobs()
    assert rc == 0 and "matrix-pipe" in route, route
    torch.cuda.synchronize()
    first = real[0][0] * real[0][1]
    assert torch.isnan(gv[:, first:]).all() and not torch.isnan(gv[:, :first]).any()


def test_matrix_pipe_scatter_replays_from_a_hip_graph(monkeypatch):
    """The backward with the matrix-pipe kernel only enqueues work (buffer resources built in the kernel, no host reads, no
    allocation): captured in a HIP graph together with the forward, its replay equals the eager call on new inputs."""
    from devis_amd import _native
    from devis_amd.functions import MSDeformAttnTemporalFunction
    monkeypatch.setenv("MSDA_SCATTER_MFMA", "1")
    T, Lq, clips = 4, 40, 2
    ds = [make_temporal_inputs(950 + c, T, T - 1, 8, 32, Lq, PYR_A, 4, 4) for c in range(clips)]
    cat = {k: (np.concatenate([x[k] for x in ds], 0) if k not in ("shapes", "lsi", "ftab") else ds[0][k]) for k in ds[0]}
    dev = lambda k: torch.from_numpy(cat[k]).to(DEV)
    shapes, lsi, ftab = dev("shapes"), dev("lsi"), dev("ftab")
    static = [dev(k).requires_grad_(True) for k in ("value", "loc_c", "aw_c", "loc_t", "aw_t")]
    go = dev("grad_out")
    _native.shapes_hint(shapes)                         # the host copy, outside the capture

    def step():
        out = MSDeformAttnTemporalFunction.apply(static[0], shapes, lsi, ftab, *static[1:], clips)
        return (out,) + torch.autograd.grad(out, static, go)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = step()
    # new inputs into the captured tensors, replay, compare with an eager call on the same inputs
    fresh = [make_temporal_inputs(970 + c, T, T - 1, 8, 32, Lq, PYR_A, 4, 4) for c in range(clips)]
    with torch.no_grad():
        for t, k in zip(static, ("value", "loc_c", "aw_c", "loc_t", "aw_t")):
            t.copy_(torch.from_numpy(np.concatenate([x[k] for x in fresh], 0)).to(DEV))
    graph.replay()
    torch.cuda.synchronize()
    got = [x.clone() for x in captured]
    want, routes = _routes_of_backward(step)
    torch.cuda.synchronize()
    assert "matrix-pipe" in routes[0]
    for i, (a, b) in enumerate(zip(got, want)):
        tol = 2e-5 if i == 1 else 1e-6                  # (grad_value: the owner kernel's list order varies from run to run)
        assert torch.allclose(a, b, rtol=0, atol=tol * max(1.0, float(b.abs().max()))), i
