"""GPU parity at the shapes BASELINE.json names beyond the headline (configs[4]: SwinL training pyramid, fp16,
im2col_step 1 vs 64, mask-head-like 3-level calls), the bench-scale batch against the ORACLE, reduced-precision
modules against the reference fixtures, the reference's largest gradcheck head dims, and the fused pre-op pass
with reference points of another dtype.  Everything goes through the C ABI; the oracle is only the checker."""
import numpy as np
import pytest
import torch

import module_cases
from conftest import golden_names
from helpers import PYR_A, make_inputs, make_temporal_inputs, oracle_fwd_bwd, round_to, temporal_reference

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# training sizes go up to 480x768 (ref src/datasets/vis.py:228-231): strides 8/16/32/64
SWIN_PYRAMID = [(60, 96), (30, 48), (15, 24), (8, 12)]
MASK_HEAD_LEVELS = [(60, 96), (30, 48), (15, 24)]       # strides /8, /16, /32 (ref src/config.py:63)


def _maxabs(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


def _run_op(d, dtype, step):
    from devis_amd.functions import MSDeformAttnFunction
    v, l, a = (torch.from_numpy(np.asarray(d[k], dtype=np.float64)).to(DEV, dtype).requires_grad_(True)
               for k in ("value", "loc", "aw"))
    shapes, lsi = torch.from_numpy(d["shapes"]).to(DEV), torch.from_numpy(d["lsi"]).to(DEV)
    out = MSDeformAttnFunction.apply(v, shapes, lsi, l, a, step)
    go = torch.from_numpy(np.asarray(d["grad_out"], dtype=np.float64)).to(DEV, dtype)
    gv, gl, ga = torch.autograd.grad(out, (v, l, a), go)
    torch.cuda.synchronize()
    return [t.detach().double().cpu().numpy() for t in (out, gv, gl, ga)]


@pytest.mark.parametrize("step", [1, 64])
@pytest.mark.parametrize("shapes,Lq", [(SWIN_PYRAMID, 300), (MASK_HEAD_LEVELS, 60)], ids=["swinl-pyramid", "mask-head-levels"])
@pytest.mark.parametrize("dtype,tol_out,tol_grad", [(torch.float16, 2e-3, 1e-2), (torch.float32, 1e-5, 1e-4)], ids=["f16", "f32"])
def test_cfg4_plain_op_swinl_shapes(shapes, Lq, step, dtype, tol_out, tol_grad):
    """BASELINE configs[4]: plain MSDeformAttn as devis_ablation_transformer_wo_t_conn.py:54-62 calls it (N = T = 6
    frames as the batch), SwinL pyramid, M = 8 x D = 32, K = 4, im2col_step 1 and 64 (ms_deform_attn_cuda.cu:50-75),
    against the fp64 oracle on the SAME rounded inputs; and the 3-level mask-head-like call (Lq = instances * T)."""
    d = make_inputs(4242, 6, 8, 32, Lq, shapes, 4, "wide", np.float32, value_scale=1.0)
    d = round_to(d, dtype)
    ref = oracle_fwd_bwd(d, np.float64)
    ref32 = oracle_fwd_bwd(d, np.float32)              # grad_loc flips sign across cell borders: same-arithmetic oracle
    got = _run_op(d, dtype, step)
    assert _maxabs(got[0], ref[0]) <= tol_out * max(1.0, np.abs(ref[0]).max())
    assert _maxabs(got[1], ref[1]) <= tol_grad * max(1.0, np.abs(ref[1]).max())
    assert _maxabs(got[3], ref[3]) <= tol_grad * max(1.0, np.abs(ref[3]).max())
    if dtype == torch.float32:
        assert _maxabs(got[2], ref32[2]) <= tol_grad * max(1.0, np.abs(ref32[2]).max())
    else:       # half: the rounded coordinates times a level's width are exact in fp32 -- the same cells as the fp64 oracle's
        assert _maxabs(got[2], ref[2]) <= tol_grad * max(1.0, np.abs(ref[2]).max())


def test_cfg4_im2col_step_values_agree_bitwise():
    """The chunk loop only moves pointers (cu:61-75): step 1 and step 64 give identical forward results -- bit for bit while the
    same kernel family serves both batch sizes (the route RULES do here); with the measured route table (devis_amd/routes.json,
    round 5) a batch of 6 may be pinned to another family than a batch of 1, and the two then agree to rounding (another order
    of the same fp32 terms), like the reference's own atomicAdd order."""
    from devis_amd import _native
    d = round_to(make_inputs(7, 6, 8, 32, 300, SWIN_PYRAMID, 4, "unit", np.float32, value_scale=1.0), torch.float16)
    _native.load()
    _native.clear_routes()
    try:
        a, b = _run_op(d, torch.float16, 1), _run_op(d, torch.float16, 64)
    finally:
        _native._load_shipped_routes()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert _maxabs(a[1], b[1]) <= 1e-3 * max(1.0, np.abs(b[1]).max())      # grad_value: summation order may differ
    a, b = _run_op(d, torch.float16, 1), _run_op(d, torch.float16, 64)      # ... and on whatever routes the table pins
    for x, y, tol in ((a[0], b[0], 2e-3), (a[1], b[1], 1e-2), (a[3], b[3], 1e-2)):
        assert _maxabs(x, y) <= tol * max(1.0, np.abs(y).max())


BENCH_SCALE = [
    # dtype, clips, value layout, tiles per wave the forward must have run with (None: not asserted), tol out, tol grads
    (torch.float32, 16, "dense", None, 1e-5, 1e-4),
    (torch.float32, 16, "padded", None, 1e-5, 1e-4),       # the layout devis_amd's own value_proj writes (modules' default)
    (torch.float32, 32, "dense", 4, 1e-5, 1e-4),           # 4 tiles per wave forced (fp32 picks 1): accumulator sets 2 and 3 in use
    (torch.bfloat16, 16, "dense", 4, 1e-2, 2e-2),          # 16-bit storage: 4 tiles per wave at the bench's 16 clips
    (torch.float16, 16, "dense", 4, 1e-3, 4e-3),
    (torch.bfloat16, 16, "loc32", 4, 1e-2, 2e-2),          # bf16 value, float32 sampling locations / weights (MSDA_BF16_LOC32)
]


@pytest.mark.parametrize("dtype,clips,layout,want_nt,tol_out,tol_grad", BENCH_SCALE,
                         ids=["f32-16", "f32-16-padded", "f32-32-nt4", "bf16-16-nt4", "f16-16-nt4", "bf16-16-fp32-sampling"])
def test_bench_scale_batch_against_the_oracle(dtype, clips, layout, want_nt, tol_out, tol_grad, monkeypatch):
    """The regimes bench.py times -- 16 / 32 cfg3 clips in ONE fused call (resident-slab forward and gather pass with
    2 or 4 tiles per wave, owner-computes scatter), fp32 and 16-bit storage, dense and padded `value` -- compared clip
    by clip with the CPU oracle in the reference's 2*T-call pattern (helpers.temporal_reference; ref semantics
    ms_deform_im2col_cuda.cuh:237-299, ms_deform_attn.py:325-364), not with other runs of the same library.  16-bit:
    the oracle runs in fp64 on the SAME rounded inputs."""
    from devis_amd import _native
    from devis_amd.functions import MSDeformAttnTemporalFunction
    T, Lq, M, D = 6, 300, 8, 32
    if want_nt is not None and dtype == torch.float32:
        monkeypatch.setenv("MSDA_FWD_RS_NT", str(want_nt))
    check = sorted({0, clips // 2 - 1, clips - 1})
    ds = [round_to(make_temporal_inputs(900 + c, T=T, W=5, M=M, D=D, Lq=Lq, shapes=PYR_A, Pc=4, Pt=4, dtype=np.float64), dtype)
          if dtype != torch.float32 else make_temporal_inputs(900 + c, T=T, W=5, M=M, D=D, Lq=Lq, shapes=PYR_A, Pc=4, Pt=4)
          for c in range(clips)]
    shapes, lsi, ftab = (torch.from_numpy(ds[0][k]).to(DEV) for k in ("shapes", "lsi", "ftab"))
    keys = ("value", "loc_c", "aw_c", "loc_t", "aw_t")
    loc32 = layout == "loc32"
    if loc32:           # only value / grad_out are rounded to the storage type; locations and weights stay fp32
        ds = [dict(make_temporal_inputs(900 + c, T=T, W=5, M=M, D=D, Lq=Lq, shapes=PYR_A, Pc=4, Pt=4),
                   **{k: round_to({k: np.asarray(v, dtype=np.float64)}, dtype)[k] for k, v in
                      make_temporal_inputs(900 + c, T=T, W=5, M=M, D=D, Lq=Lq, shapes=PYR_A, Pc=4, Pt=4).items() if k in ("value", "grad_out")})
              for c in range(clips)]
    cat = lambda k: torch.from_numpy(np.concatenate([d[k] for d in ds], 0)).to(
        DEV, torch.float32 if (loc32 and k in ("loc_c", "aw_c", "loc_t", "aw_t")) else dtype)
    leaves = [cat(k).requires_grad_(True) for k in keys]
    if layout == "padded":
        buf = torch.zeros((clips * T, leaves[0].shape[1], M + 1, D), dtype=dtype, device=DEV)
        buf[:, :, :M] = leaves[0].detach()
        leaves[0] = buf[:, :, :M].requires_grad_(True)
        assert not leaves[0].is_contiguous()
    go = cat("grad_out")
    routes = []
    leaves[0].register_hook(lambda g: routes.append(_native.last_route()))      # (the backward runs on autograd's thread)
    out = MSDeformAttnTemporalFunction.apply(leaves[0], shapes, lsi, ftab, *leaves[1:], clips)
    fwd_route = _native.last_route()
    grads = torch.autograd.grad(out, leaves, go)
    torch.cuda.synchronize()
    assert "resident-slab" in fwd_route and "resident-slab" in routes[0] and "owner-computes" in routes[0], (fwd_route, routes)
    if want_nt is not None:
        assert "%d tiles per wave" % want_nt in fwd_route, fwd_route
    got = [x.detach().double().cpu().numpy() for x in (out,) + tuple(grads)]
    fkeys = ("value", "shapes", "lsi", "ftab", "loc_c", "aw_c", "loc_t", "aw_t", "grad_out")
    for c in check:
        d = ds[c]
        ref = temporal_reference(*[np.asarray(d[k], dtype=np.float64) if d[k].dtype.kind == "f" else d[k] for k in fkeys])
        # fp32 oracle for grad_loc (a location within rounding distance of a cell border takes the other cell's derivative)
        ref32 = temporal_reference(*[np.asarray(d[k], dtype=np.float32) if d[k].dtype.kind == "f" else d[k] for k in fkeys])
        names = ("out", "grad_value", "grad_loc_c", "grad_aw_c", "grad_loc_t", "grad_aw_t")
        for i, name in enumerate(names):
            mine = got[i][c * T:(c + 1) * T]
            want = ref32[i] if name.startswith("grad_loc") else ref[i]
            tol = tol_grad if i else tol_out
            scale = max(1.0, np.abs(want).max())
            if name.startswith("grad_loc") and dtype != torch.float32 and not loc32:
                want = ref[i]       # 16-bit coordinates x a level's width are exact in fp32: the fp64 oracle's cells
            assert _maxabs(mine, want) <= tol * scale, (c, name, _maxabs(mine, want), scale)


MODULE_FIXTURES = [n for n in golden_names("mod_") if n != "mod_fresh_init"]


@pytest.mark.parametrize("dtype,sampling,tol_out,tol_grad", [
    (torch.bfloat16, "fp32", 1e-2, 1e-1), (torch.float16, "fp32", 2e-3, 3e-2),
    (torch.bfloat16, "storage", 1e-1, 5e-1), (torch.float16, "storage", 1.5e-2, 1.5e-1)],
    ids=["bf16-fp32-sampling", "f16-fp32-sampling", "bf16", "f16"])
@pytest.mark.parametrize("kind", ["dec", "enc"])
def test_config_sized_modules_reduced_precision_vs_reference_fixture(kind, dtype, sampling, tol_out, tol_grad):
    """The temporal decoder / encoder at DeVIS's real size in bf16 / f16 (parameters, activations and `value` stored in
    16 bits, arithmetic fp32) against the fp64 fixture captured from the REFERENCE modules (tests/golden/cfg_*.npz),
    norm-wise over the sampled elements.
    * `fp32` sampling (the modules' default, ABI v11 MSDA_*_LOC32): the fused pre-op pass hands sampling locations and
      attention weights to the operator in float32 and uses the (fp32) reference points unrounded -- outputs within the
      north_star's 1e-2 for bf16.
    * `storage` sampling (`sampling_fp32 = False`, reference points in the module's dtype): everything in 16 bits; a
      normalised coordinate in [0.5, 1) then resolves 2^-9 in bf16 = 0.16 px on the 80-pixel-wide level 0 (f16: 0.02 px) on
      white-noise maps: measured 6e-2 (bf16) / 9e-3 (f16) on outputs.
    Gradients are piecewise-constant bilinear derivatives -- a location rounded across a cell border flips single
    terms -- hence the looser norm-wise bound on them."""
    from conftest import golden
    from devis_amd.modules import TemporalMSDeformAttnDecoder, TemporalMSDeformAttnEncoder
    g = golden("cfg_" + kind)
    cls = TemporalMSDeformAttnDecoder if kind == "dec" else TemporalMSDeformAttnEncoder
    got = module_cases.cfg_run(kind, cls, device=DEV, dtype=dtype, ref_dtype=torch.float32 if sampling == "fp32" else None,
                               sampling_fp32=sampling == "fp32")
    report = {}
    for k, v in got.items():
        assert np.isfinite(v).all(), k
        mine, want = module_cases.cfg_sample(v)["sample"], g[k + "/sample"]
        report[k] = float(np.linalg.norm(mine - want) / max(1e-12, np.linalg.norm(want)))
    print("cfg_%s %s %s-sampling relative errors:" % (kind, dtype, sampling), {k: "%.2e" % e for k, e in report.items()})
    for k, rel in report.items():
        assert rel <= (tol_out if (k == "out" or k.startswith("aux/")) else tol_grad), (k, rel)


def test_gradcheck_reference_large_head_dims():
    """The reference's own gradcheck procedure (src/models/ops/test.py:19-35,83) at its largest head dims
    D = 2048 and 3096 (round 1 stopped at 1025)."""
    from devis_amd.functions import MSDeformAttnFunction
    for D in (2048, 3096):
        torch.manual_seed(3)
        N, M, Lq, L, P = 1, 2, 2, 2, 2
        shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long, device=DEV)
        lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        S = int(shapes.prod(1).sum())
        value = (torch.rand(N, S, M, D, device=DEV) * 0.01).double().requires_grad_(True)
        loc = torch.rand(N, Lq, M, L, P, 2, device=DEV).double().requires_grad_(True)
        aw = torch.rand(N, Lq, M, L, P, device=DEV) + 1e-5
        aw = (aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)).double().requires_grad_(True)
        assert torch.autograd.gradcheck(MSDeformAttnFunction.apply, (value, shapes, lsi, loc, aw, 2))


def test_fused_prep_takes_reference_points_of_another_dtype():
    """DeVIS builds reference points in fp32 (get_reference_points) whatever the model's dtype; the fused pre-op pass
    reads raw pointers, so it must cast rather than reinterpret: a bf16 module fed fp32 reference points
    with 16-bit sampling (`sampling_fp32 = False`) equals the same module fed bf16 reference points, and agrees with the
    torch-op path; with fp32 sampling (the default) the fp32 reference points are used unrounded."""
    from devis_amd.modules import MSDeformAttn
    torch.manual_seed(1)
    C, M, L, P, N, Lq = 64, 8, 2, 4, 2, 50
    shapes = torch.tensor([(12, 10), (6, 5)], device=DEV)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    mod = MSDeformAttn(C, L, M, P).to(DEV)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn_like(p) * 0.1)
    mod = mod.to(torch.bfloat16)
    query = torch.randn(N, Lq, C, device=DEV, dtype=torch.bfloat16)
    src = torch.randn(N, S, C, device=DEV, dtype=torch.bfloat16)
    ref32 = torch.rand(N, Lq, L, 2, device=DEV, dtype=torch.float32)
    ref16 = ref32.to(torch.bfloat16)
    out_f = mod(query, ref32, src, shapes, lsi, None)[0]            # fp32 sampling: unrounded reference points
    mod.sampling_fp32 = False
    out_a = mod(query, ref32, src, shapes, lsi, None)[0]
    out_b = mod(query, ref16, src, shapes, lsi, None)[0]
    assert torch.isfinite(out_a).all() and torch.equal(out_a, out_b)
    assert torch.isfinite(out_f).all() and not torch.equal(out_f, out_a)
    assert (out_f.float() - out_a.float()).abs().max().item() <= 5e-2 * max(1.0, out_a.float().abs().max().item())
    mod.fused_prep = False
    out_c = mod(query, ref16, src, shapes, lsi, None)[0]
    assert (out_a.float() - out_c.float()).abs().max().item() <= 5e-2 * max(1.0, out_c.float().abs().max().item())
    # shapes on the wrong device / of the wrong dtype fail loudly instead of being dereferenced
    mod.fused_prep = True
    with pytest.raises(RuntimeError):
        mod(query, ref32, src, shapes.to(torch.int32), lsi, None)


def test_frame_table_wraps_negative_offsets_on_the_device():
    """temporal_offsets[t] + t indexes `value` with Python semantics in the reference (ms_deform_attn.py:339,445):
    a negative index wraps once.  The table is built without a host synchronisation (out-of-range offsets: the next
    test)."""
    from devis_amd.modules.ms_deform_attn import TemporalMSDeformAttnBase
    T = 4
    offs = [torch.tensor([-1, 1], device=DEV) for _ in range(T)]
    offs[0] = torch.tensor([-1, 1], device=DEV)        # frame 0 - 1 -> wraps to T-1
    offs[T - 1] = torch.tensor([-1, -2], device=DEV)
    table = TemporalMSDeformAttnBase._frame_table(offs, T, torch.device(DEV)).cpu().tolist()
    assert table[0] == [T - 1, 1] and table[T - 1] == [T - 2, T - 3]


def test_frame_table_rejects_out_of_range_offsets_on_the_device_without_a_sync():
    """Device-side offsets outside the clip (the reference's indexing kernel trips a device-side assert there): the
    table is still built without waiting -- wrapped into range, so nothing outside the clip is ever read -- and the
    IndexError surfaces at the first later call that finds the verdict on the host; the process survives."""
    from devis_amd.modules.ms_deform_attn import TemporalMSDeformAttnBase as B
    T = 3
    B._raise_on_bad_offsets(wait=True)                  # (nothing pending from other tests)
    bad = [torch.tensor([1, 7], device=DEV) for _ in range(T)]
    table = B._frame_table(bad, T, torch.device(DEV))
    assert int(table.min()) >= 0 and int(table.max()) < T
    with pytest.raises(IndexError, match="outside the clip"):
        B._raise_on_bad_offsets(wait=True)
    good = [torch.tensor([o for o in range(-f, T - f) if o != 0], device=DEV) for f in range(T)]       # connect-all (devis_transformer.py:103-105)
    assert B._frame_table(good, T, torch.device(DEV)).shape == (T, 2)      # the module keeps working
    B._raise_on_bad_offsets(wait=True)


def test_strict_temporal_offsets_raise_at_once_and_name_the_offsets(monkeypatch):
    """``STRICT_TEMPORAL_OFFSETS``: one synchronisation per NEW list of device-side offsets buys the reference's behaviour
    (its indexing assert fires in the call that made the mistake); the message carries the offending offsets."""
    import devis_amd.modules.ms_deform_attn as mm
    mm.TemporalMSDeformAttnBase._raise_on_bad_offsets(wait=True)
    monkeypatch.setattr(mm, "STRICT_TEMPORAL_OFFSETS", True)
    bad = [torch.tensor([1, 7], device=DEV) for _ in range(3)]
    with pytest.raises(IndexError, match=r"temporal_offsets = \[\[1, 7\]"):
        mm.TemporalMSDeformAttnBase._frame_table(bad, 3, torch.device(DEV))
    mm.TemporalMSDeformAttnBase._raise_on_bad_offsets(wait=True)      # nothing was left pending


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 1e-4)], ids=["f64", "f32"])
@pytest.mark.parametrize("kind", ["dec", "enc"])
def test_config_sized_modules_vs_reference_fixture(kind, dtype, tol):
    """SURVEY 8c harness row: the temporal decoder / encoder at DeVIS's real size (C=256, M=8, L=4, T=6, 300 queries
    per frame resp. Lq = S = 4820, 360x640 pyramid) against the fixture captured from the REFERENCE modules
    (tests/golden/cfg_*.npz: statistics + a strided subsample of every output and gradient; parameters and inputs
    are regenerated from the same seeds).  fp64 pins the arithmetic, fp32 the BASELINE bar of 1e-4."""
    from conftest import golden
    from devis_amd.modules import TemporalMSDeformAttnDecoder, TemporalMSDeformAttnEncoder
    g = golden("cfg_" + kind)
    cls = TemporalMSDeformAttnDecoder if kind == "dec" else TemporalMSDeformAttnEncoder
    got = module_cases.cfg_run(kind, cls, device=DEV, dtype=dtype)
    assert sorted(k + "/sample" for k in got) == sorted(k for k in g if k.endswith("/sample"))
    for k, v in got.items():
        mine = module_cases.cfg_sample(v)
        want_s, want_t = g[k + "/sample"], g[k + "/stats"]
        scale = max(1.0, float(np.abs(want_s).max()), abs(float(want_t[3])), abs(float(want_t[4])))
        assert mine["sample"].shape == want_s.shape, k
        if dtype == torch.float32 and k.startswith("grad/"):
            # fp32 gradients: a location within rounding distance of a cell border takes the neighbouring cell's
            # (piecewise constant) derivative -- isolated terms differ at 1e-3: a norm-wise bound
            err = np.abs(mine["sample"] - want_s)
            assert float(np.linalg.norm(err) / max(1e-12, np.linalg.norm(want_s))) <= 5e-3, k
            continue
        assert float(np.abs(mine["sample"] - want_s).max()) <= tol * scale, (k, float(np.abs(mine["sample"] - want_s).max()), scale)
        # sums over the WHOLE tensor: catches an error anywhere, not only at the sampled positions
        n = v.size
        assert abs(mine["stats"][0] - want_t[0]) <= tol * scale * n ** 0.5 * 4 + tol * abs(want_t[0]), (k, "sum")
        assert abs(mine["stats"][1] - want_t[1]) <= tol * (want_t[1] + scale), (k, "abs-sum")
        assert abs(mine["stats"][2] - want_t[2]) <= 4 * tol * (want_t[2] + scale), (k, "square-sum")


def _run_temporal(d, dtype, clips, routes):
    from devis_amd import _native
    from devis_amd.functions import MSDeformAttnTemporalFunction
    f = lambda k: torch.from_numpy(np.asarray(d[k], dtype=np.float64)).to(DEV, dtype).requires_grad_(True)
    v, lc, ac, lt, at = f("value"), f("loc_c"), f("aw_c"), f("loc_t"), f("aw_t")
    # msda_last_route is per thread and the backward runs on autograd's device thread: read it from a hook there
    v.register_hook(lambda g: routes.append(_native.last_route()))
    shapes, lsi, ftab = (torch.from_numpy(d[k]).to(DEV) for k in ("shapes", "lsi", "ftab"))
    out = MSDeformAttnTemporalFunction.apply(v, shapes, lsi, ftab, lc, ac, lt, at, clips)
    go = torch.from_numpy(np.asarray(d["grad_out"], dtype=np.float64)).to(DEV, dtype)
    grads = torch.autograd.grad(out, (v, lc, ac, lt, at), go)
    torch.cuda.synchronize()
    return [t.detach().double().cpu().numpy() for t in (out,) + tuple(grads)]


RS_CASES = [
    # T, W, frame table (None = every other frame), Lq, pyramid, Pc, Pt
    (5, 2, [[1, 1], [0, 2], [1, 3], [2, 4], [3, 3]], 37, [(6, 5), (3, 3)], 3, 2),       # mirrored window, odd points
    (3, 2, None, 16, [(9, 7), (5, 4), (3, 2), (2, 1)], 4, 4),                            # 4 levels, tile-exact rows
    (4, 3, None, 50, [(12, 10)], 1, 1),                                                  # one level, one point
    (2, 1, [[1], [1]], 19, [(7, 6), (4, 3), (2, 2)], 2, 3),                              # a frame nobody else reads
]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2), (torch.float16, 2e-3)], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", range(len(RS_CASES)))
def test_round2_kernels_forced_on_odd_shapes(case, dtype, tol, monkeypatch):
    """The resident-slab forward / gather pass and the owner-computes scatter, FORCED onto shapes they would not pick by
    themselves (few rows, 1-3 points per level, 1-4 levels, mirrored / partial frame tables, 2 clips), against the fp64
    oracle in the reference's call pattern on the same rounded inputs; msda_last_route confirms the kernels ran."""
    from devis_amd import _native
    T, W, ftab, Lq, shapes, Pc, Pt = RS_CASES[case]
    monkeypatch.setenv("MSDA_FWD_RS", "1"); monkeypatch.setenv("MSDA_BWD_RS", "1")
    clips = 2
    ft = None if ftab is None else np.array(ftab, dtype=np.int32)
    ds = [round_to(make_temporal_inputs(500 + 10 * case + c, T, W, 8, 32, Lq, shapes, Pc, Pt, ftab=ft), dtype) for c in range(clips)]
    cat = {k: (np.concatenate([x[k] for x in ds], 0) if k not in ("shapes", "lsi", "ftab") else ds[0][k]) for k in ds[0]}
    routes = []
    got = _run_temporal(cat, dtype, clips, routes)
    route = _native.last_route() + " | " + " | ".join(routes)
    assert "forward (resident-slab" in route and "resident-slab kernel, grad_loc" in route and "owner-computes" in route, route
    for c, d in enumerate(ds):
        args = [np.asarray(d[k], dtype=np.float64) if d[k].dtype.kind == "f" else d[k]
                for k in ("value", "shapes", "lsi", "ftab", "loc_c", "aw_c", "loc_t", "aw_t", "grad_out")]
        ref = temporal_reference(*args)
        for i, (a, b) in enumerate(zip(got, ref)):
            mine = a[c * T:(c + 1) * T]
            if i in (2, 4) and dtype == torch.float32:         # grad_loc in fp32: cell borders (see the other tests) -- all but a sliver of the entries
                bad = np.abs(mine - b) > 10 * tol * max(1.0, np.abs(b).max())
                assert bad.mean() <= 1e-3, (c, i, bad.mean())
            else:
                assert _maxabs(mine, b) <= tol * max(1.0, np.abs(b).max()), (c, i)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16, torch.float16], ids=["f32", "f64", "bf16", "f16"])
def test_mask_rows_zeroes_exactly_the_masked_rows(dtype):
    """msda_mask_rows (SURVEY f-3; ref ms_deform_attn.py:102-103): dense and row-padded tensors, 16-byte and odd row
    sizes, an unaligned base; equal to masked_fill bit for bit, pad columns and unmasked rows untouched (NaN canaries)."""
    from devis_amd import _native
    g = torch.Generator().manual_seed(11)
    for pixels, elems, stride, offset in ((1000, 256, 256, 0), (777, 256, 288, 0), (130, 20, 27, 0), (64, 24, 24, 3), (1, 8, 8, 0)):
        buf = torch.randn(pixels * stride + offset + 8, generator=g).to(DEV, dtype)
        rows = buf[offset:offset + pixels * stride].view(pixels, stride)
        rows[:, elems:] = float("nan")                                  # the pad columns must not be touched
        mask = (torch.rand(pixels, generator=g) < 0.3).to(DEV)
        want = rows.clone()
        want[:, :elems] = want[:, :elems].masked_fill(mask[:, None], 0.0)
        _native.mask_rows(rows[:, :elems] if stride > elems else rows, mask, elems)
        torch.cuda.synchronize()
        assert torch.equal(torch.nan_to_num(rows, nan=123.0), torch.nan_to_num(want, nan=123.0)), (pixels, elems, stride)
        assert torch.isnan(rows[:, elems:]).all()
    with pytest.raises(ValueError):
        _native.mask_rows(rows, mask[:0], 8)


@pytest.mark.parametrize("pad", [0, 1])
def test_plain_module_mask_is_applied_by_mask_rows_both_ways(pad, monkeypatch):
    """MSDeformAttn with a padding mask: value is masked by msda_mask_rows and grad_value by the operator's backward
    (no masked_fill pass over either); same numbers as the reference module fixtures (which carry a mask), in the
    padded and the dense value layout."""
    from devis_amd import _native
    from devis_amd.modules import MSDeformAttn
    monkeypatch.setattr(MSDeformAttn, "value_pad_heads", pad)
    calls = []
    real = _native.mask_rows
    monkeypatch.setattr(_native, "mask_rows", lambda *a: (calls.append(a[0].shape), real(*a))[1])
    got, g = module_cases.run("mod_plain_ref2", DEV, torch.float64)
    module_cases.compare(got, g, rtol=1e-9, atol=1e-11)
    assert len(calls) == 2, calls                                      # value, then grad_value
    got, g = module_cases.run("mod_plain_ref4", DEV, torch.float32)
    module_cases.compare(got, g, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["f32", "f16"])
@pytest.mark.parametrize("env", [{}, {"MSDA_FWD_RS": "1", "MSDA_BWD_RS": "1"}, {"MSDA_FWD_RS": "0", "MSDA_BWD_RS": "0"},
                                 {"MSDA_FORCE_GENERIC": "1"}, {"MSDA_SCATTER_OWN": "0"}, {"MSDA_BWD_MODE": "atomic"}],
                         ids=["default", "resident-slab", "tile", "generic", "lds-scatter", "atomic"])
def test_non_finite_pixels_nobody_samples_do_not_leak(env, dtype, monkeypatch):
    """The reference never reads a corner outside the map (cuh:56-80, 265-288), so an inf / NaN in a pixel no point
    samples cannot reach any output.  Kernels that fetch a placeholder for such corners must fetch zeros, not
    pixel 0 times weight 0.  Here the top-left pixel of EVERY level of every frame is inf or NaN, all
    points sit in the lower-right part of the maps or beyond their edge (many corners outside), fused op, every
    forward / backward route: outputs and gradients are finite and equal the oracle's on the same tensors with
    those pixels set to 0."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    T, W, shapes = 3, 2, [(9, 8), (5, 4), (3, 3)]
    d = make_temporal_inputs(77, T, W, 8, 32, 41, shapes, 4, 2)
    for k in ("loc_c", "loc_t"):
        d[k] = (0.7 + 0.55 * np.random.default_rng(5).random(d[k].shape)).astype(d[k].dtype)      # [0.7, 1.25)
    d = round_to(d, dtype)
    clean = dict(d)
    poisoned = d["value"].copy()
    for i, s in enumerate(d["lsi"]):
        poisoned[:, s] = np.inf if i % 2 == 0 else np.nan
    ref = temporal_reference(*(np.asarray(clean[k], dtype=np.float64) if clean[k].dtype.kind == "f" else clean[k]
                               for k in ("value", "shapes", "lsi", "ftab", "loc_c", "aw_c", "loc_t", "aw_t", "grad_out")))
    z = clean["value"].copy()
    z[:, d["lsi"]] = 0.0
    ref0 = temporal_reference(*(np.asarray(x, dtype=np.float64) if x.dtype.kind == "f" else x
                                for x in (z,) + tuple(clean[k] for k in ("shapes", "lsi", "ftab", "loc_c", "aw_c", "loc_t", "aw_t", "grad_out"))))
    assert all(np.array_equal(a, b) for a, b in zip(ref[:1] + ref[2:], ref0[:1] + ref0[2:]))      # nobody samples those pixels
    d["value"] = poisoned
    got = _run_temporal(d, dtype, 1, [])
    tol = 2e-5 if dtype == torch.float32 else 2e-3
    for i, (a, b) in enumerate(zip(got, ref)):
        assert np.isfinite(a).all(), (i, np.argwhere(~np.isfinite(a))[:4])
        if i in (2, 4) and dtype != torch.float32:
            continue                                                     # grad_loc at 16-bit: cell borders, tested elsewhere
        assert _maxabs(a, b) <= tol * max(1.0, np.abs(b).max()), i


@pytest.mark.parametrize("dtype,tol,tol_loc", [(torch.float32, 2e-5, 2e-4), (torch.float16, 2e-3, 1e-2)], ids=["f32", "f16"])
@pytest.mark.parametrize("Lq", [60, 180, 7], ids=["yt-vis-60", "ovis-180", "q7"])
def test_one_clip_at_the_query_counts_of_the_shipped_configs(Lq, dtype, tol, tol_loc):
    """DeVIS's shipped configs run 60 (YouTube-VIS) / 180 (OVIS) queries per frame, not the code's default of 300:
    one clip, T = 6, connect-all, on the 360x640 pyramid, against the fp64 oracle on the same rounded inputs.  The gather pass of such
    a call takes the resident-slab kernel with one source frame per workgroup (round 4) -- also with 7 queries, a fraction of a tile."""
    T = 6
    d = round_to(make_temporal_inputs(2100 + Lq, T=T, W=T - 1, M=8, D=32, Lq=Lq, shapes=PYR_A, Pc=4, Pt=4, dtype=np.float64), dtype)
    routes = []
    got = _run_temporal(d, dtype, 1, routes)
    assert "one source frame per workgroup" in routes[0] and "owner-computes" in routes[0], routes
    keys = ("value", "shapes", "lsi", "ftab", "loc_c", "aw_c", "loc_t", "aw_t", "grad_out")
    ref = temporal_reference(*[np.asarray(d[k], dtype=np.float64) if d[k].dtype.kind == "f" else d[k] for k in keys])
    ref32 = temporal_reference(*[np.asarray(d[k], dtype=np.float32) if d[k].dtype.kind == "f" else d[k] for k in keys]) \
        if dtype == torch.float32 else ref
    for i, (a, b) in enumerate(zip(got, ref)):
        want = ref32[i] if i in (2, 4) else b              # fp32 grad_loc: same-arithmetic oracle (cell borders)
        bound = (tol_loc if i in (2, 4) else tol) * max(1.0, np.abs(want).max())
        assert _maxabs(a, want) <= bound, (i, _maxabs(a, want), bound)
