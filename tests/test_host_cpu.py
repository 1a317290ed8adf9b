"""CPU tests of the host side: C-ABI library (builds, loads, exports every symbol of include/msda.h,
argument errors -- no compute calls), operator wiring and the nn.Modules against the golden fixtures
made from the REFERENCE modules.  The kernels are replaced by an oracle-backed test double
(tests/fake_native.py) -- this checks the Python, not the HIP code (that is tests -m gpu)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import fake_native
import module_cases
from conftest import ROOT, golden, golden_names

MODULE_FIXTURES = [n for n in golden_names("mod_") if n != "mod_fresh_init"]


def test_library_builds_loads_and_exports_every_declared_symbol():
    from devis_amd import _native, build
    path = build.build()
    assert os.path.exists(path)
    lib = _native.load()
    header = open(os.path.join(ROOT, "include", "msda.h")).read()
    declared = set(re.findall(r"\b(msda_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_native.EXPORTED_SYMBOLS)
    raw = ctypes.CDLL(path)
    for name in declared:
        assert hasattr(raw, name), name
    assert lib.msda_version() == int(re.search(r"#define MSDA_ABI_VERSION (\d+)", header).group(1))
    assert _native.BWD_WORKSPACE_BYTES == int(re.search(r"#define MSDA_BWD_WORKSPACE_BYTES (\d+)", header).group(1))


def test_build_info_and_timing_only_guards(tmp_path, monkeypatch):
    """ABI v13 hygiene: the shipped library is not a timing-only build and says so; the timing-only
    experiment macros do not compile without -DMSDA_TIMING_ONLY_BUILD; MSDA_LIB is an error without MSDA_ENABLE_HOOKS=1."""
    import shutil
    import subprocess
    from devis_amd import _native, build
    lib = _native.load()
    info = lib.msda_build_info().decode()
    assert "abi=%d" % _native.MSDA_ABI_VERSION in info and "gfx950" in info and "timing_only=0" in info
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if os.path.exists(hipcc):
        probe = tmp_path / "probe.hip"
        probe.write_text('#include "msda_common.h"\n')
        base = [hipcc, "--offload-arch=gfx950", "-E", "-I", os.path.join(ROOT, "include"), "-I", build.CSRC, str(probe), "-o", os.devnull]
        for macro in ("-DMSDA_RS_EXP=6", "-DMSDA_WIN_EXP=1", "-DMSDA_MFMA_EXP=5"):
            r = subprocess.run(base + [macro], capture_output=True, text=True)
            assert r.returncode != 0 and "MSDA_TIMING_ONLY_BUILD" in r.stderr, (macro, r.stderr[-400:])
            r = subprocess.run(base + [macro, "-DMSDA_TIMING_ONLY_BUILD"], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-400:]
    monkeypatch.setenv("MSDA_LIB", build.LIB)
    monkeypatch.delenv("MSDA_ENABLE_HOOKS", raising=False)
    with pytest.raises(RuntimeError, match="MSDA_ENABLE_HOOKS"):
        build.lib_path()
    monkeypatch.setenv("MSDA_ENABLE_HOOKS", "1")
    assert build.lib_path() == build.LIB and build.ensure() == build.LIB


def test_routes_file_of_another_architecture_is_skipped(tmp_path, monkeypatch):
    import json
    from devis_amd import _native
    _native.load()
    before = _native.route_count()
    key = _native.route_key(False, 0, 3, 6, 5, 4820, 8, 32, 4, 77, 4, 4, [[45, 80], [23, 40], [12, 20], [6, 10]])
    path = tmp_path / "routes.json"
    path.write_text(json.dumps({"device": "something else (gfx942)", "routes": {key: {"fwd_rs_nt": 1}}}))
    monkeypatch.setattr(_native, "_running_arch", lambda: "gfx950")
    with pytest.warns(UserWarning, match="gfx942"):
        assert _native.load_routes(str(path)) == 0
    assert _native.route_count() == before
    path.write_text(json.dumps({"device": "MI355X (gfx950)", "routes": {key: {"fwd_rs_nt": 1}}}))
    try:
        assert _native.load_routes(str(path)) == 1 and _native.route_count() == before + 1
    finally:
        _native.pin_route(key, "")


def test_abi_argument_errors_without_gpu():
    from devis_amd import _native
    lib = _native.load()
    # null pointers / bad sizes are rejected before any HIP call
    rc = lib.msda_forward(0, None, None, None, None, None, 1, 30, 2, 2, 2, 2, 2, None, None, None, None)
    assert rc == -1 and b"null pointer" in lib.msda_last_error()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.msda_forward(0, p, p, p, p, p, 1, 30, 0, 2, 2, 2, 2, p, None, None, None)
    assert rc == -1 and b"positive" in lib.msda_last_error()
    assert lib.msda_forward(0, p, p, p, p, p, 0, 30, 2, 2, 2, 2, 2, p, None, None, None) == 0      # empty batch: no-op
    rc = lib.msda_temporal_forward(0, p, p, p, p, p, p, p, p, 1, 0, 1, 30, 2, 2, 2, 2, 2, 2, p, None, None, None)
    assert rc == -1


def test_operator_raises_on_cpu_tensors():
    """ms_deform_attn.h:38,60: 'Not implemented on the CPU' -- and no silent fallback."""
    from devis_amd.functions import MSDeformAttnFunction
    g = golden("op_testpy_shape")
    args = [torch.from_numpy(g[k]) for k in ("value", "spatial_shapes", "level_start_index",
                                             "sampling_locations", "attention_weights")]
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDeformAttnFunction.apply(*args, 2)


def test_debug_core_pytorch_matches_reference():
    """ms_deform_attn_core_pytorch (debug helper kept in the import surface) vs golden, fp64."""
    from devis_amd.functions import ms_deform_attn_core_pytorch
    for name in ("op_testpy_shape", "op_out_of_range", "op_devis_small", "op_generic_D71"):
        g = golden(name)
        out = ms_deform_attn_core_pytorch(torch.from_numpy(g["value"]).double(), torch.from_numpy(g["spatial_shapes"]),
                                          torch.from_numpy(g["sampling_locations"]).double(),
                                          torch.from_numpy(g["attention_weights"]).double())
        np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-11, atol=1e-14)


def test_function_wiring_and_im2col_chunking(monkeypatch):
    """autograd contract (grads for args 0,3,4 only), chunk loop, divisibility error."""
    fake_native.install(monkeypatch)
    from devis_amd.functions import MSDeformAttnFunction
    g = golden("op_batched_im2col")
    for step in (1, 2, 3, 6, 64):
        v, l, a = (torch.from_numpy(g[k]).double().requires_grad_(True)
                   for k in ("value", "sampling_locations", "attention_weights"))
        out = MSDeformAttnFunction.apply(v, torch.from_numpy(g["spatial_shapes"]),
                                         torch.from_numpy(g["level_start_index"]), l, a, step)
        np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-12, atol=1e-15)
        gv, gl, ga = torch.autograd.grad(out, (v, l, a), torch.from_numpy(g["grad_output"]))
        np.testing.assert_allclose(gv.numpy(), g["grad_value"], rtol=1e-11, atol=1e-14)
        np.testing.assert_allclose(gl.numpy(), g["grad_sampling_loc"], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(ga.numpy(), g["grad_attn_weight"], rtol=1e-11, atol=1e-14)
    with pytest.raises(RuntimeError, match="must divide"):
        MSDeformAttnFunction.apply(v, torch.from_numpy(g["spatial_shapes"]),
                                   torch.from_numpy(g["level_start_index"]), l, a, 4)
    with pytest.raises(RuntimeError, match="contiguous"):
        MSDeformAttnFunction.apply(v.transpose(2, 3).contiguous().transpose(2, 3), torch.from_numpy(g["spatial_shapes"]),
                                   torch.from_numpy(g["level_start_index"]), l, a, 2)


def test_function_accepts_an_explicit_none_padding_mask(monkeypatch):
    """apply(v, ss, lsi, loc, aw, step, None) -- seven inputs, the optional mask explicitly None -- and the
    reference's six-input call both get one gradient slot per input."""
    fake_native.install(monkeypatch)
    from devis_amd.functions import MSDeformAttnFunction
    g = golden("op_testpy_shape")
    ss, lsi = torch.from_numpy(g["spatial_shapes"]), torch.from_numpy(g["level_start_index"])
    for extra in ((), (None,)):
        v, l, a = (torch.from_numpy(g[k]).double().requires_grad_(True)
                   for k in ("value", "sampling_locations", "attention_weights"))
        out = MSDeformAttnFunction.apply(v, ss, lsi, l, a, 2, *extra)
        gv, gl, ga = torch.autograd.grad(out, (v, l, a), torch.from_numpy(g["grad_output"]))
        np.testing.assert_allclose(gv.numpy(), g["grad_value"], rtol=1e-11, atol=1e-14)
        np.testing.assert_allclose(ga.numpy(), g["grad_attn_weight"], rtol=1e-11, atol=1e-14)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", MODULE_FIXTURES)
def test_modules_match_reference_modules(monkeypatch, name, fused):
    """Same state_dict + inputs -> same outputs, aux returns and gradients as the reference modules
    (fp64; kernels replaced by the oracle).  fused=False replays the reference's 2*T-call pattern."""
    if name.startswith("mod_plain") and not fused:
        pytest.skip("plain module has a single call pattern")
    fake_native.install(monkeypatch)
    got, g = module_cases.run(name, "cpu", torch.float64, fused=fused)
    module_cases.compare(got, g, rtol=1e-9, atol=1e-11)


def test_fresh_initialisation_matches_reference():
    """_reset_parameters (ms_deform_attn.py:64-82, 169-213): deterministic parts of the init."""
    from devis_amd.modules import MSDeformAttn, TemporalMSDeformAttnEncoder
    g = golden("mod_fresh_init")
    mods = {"plain": MSDeformAttn(32, 2, 4, 3), "temporal": TemporalMSDeformAttnEncoder(3, 32, 2, 2, 4, 3, 2)}
    seen = 0
    for key, exp in g.items():
        which, name = key.split("/", 1)
        np.testing.assert_allclose(mods[which].state_dict()[name].numpy(), exp, rtol=0, atol=1e-7)
        seen += 1
    assert seen >= 12


def test_constructor_contract():
    from devis_amd.modules import MSDeformAttn, TemporalMSDeformAttnDecoder
    with pytest.raises(ValueError):
        MSDeformAttn(d_model=30, n_heads=4)                      # ms_deform_attn.py:40-42
    with pytest.warns(UserWarning):
        MSDeformAttn(d_model=24, n_heads=4)                      # :45-48 (D = 6 not a power of 2)
    m = TemporalMSDeformAttnDecoder()
    assert (m.im2col_step, m.n_frames, m.t_window, m.n_curr_points, m.n_temporal_points) == (64, 36, 2, 4, 2)
    assert m.dec_instance_aware_att is True
    with pytest.raises(ValueError):
        m_plain = MSDeformAttn(32, 2, 4, 3)
        m_plain(torch.zeros(1, 2, 32), torch.zeros(1, 2, 2, 3), torch.zeros(1, 30, 32),
                torch.tensor([[6, 4], [3, 2]]), torch.tensor([0, 24]), None)
    with pytest.raises(AssertionError):                          # :96: sum of H*W against the length of input_flatten
        m_plain(torch.zeros(1, 2, 32), torch.zeros(1, 2, 2, 2), torch.zeros(1, 29, 32),
                torch.tensor([[6, 4], [3, 2]]), torch.tensor([0, 24]), None)


def test_frame_table_is_cached_per_offset_tensors():
    """one table per list of offset tensors, rebuilt when they change."""
    from devis_amd.modules import TemporalMSDeformAttnDecoder as Dec
    offs = [torch.tensor([1, 2]), torch.tensor([-1, 1]), torch.tensor([-2, -1])]
    a = Dec._frame_table(offs, 3, torch.device("cpu"))
    assert a.dtype == torch.int32 and a.tolist() == [[1, 2], [0, 2], [0, 1]]
    assert Dec._frame_table(list(offs), 3, torch.device("cpu")) is a          # same tensors, new list: hit
    offs[1].add_(0)                                                            # in-place change: version bump
    b = Dec._frame_table(offs, 3, torch.device("cpu"))
    assert b is not a and b.tolist() == a.tolist()
    c = Dec._frame_table([o.clone() for o in offs], 3, torch.device("cpu"))    # other tensors: miss
    assert c is not b


def test_route_table_keys_pins_and_files(tmp_path):
    """include/msda.h ABI v12: the measured route table.  Keys come from the library (one definition), pins are added, replaced,
    removed and counted, malformed settings are refused, and a routes file (the format of devis_amd/routes.json) loads."""
    import json
    from devis_amd import _native
    _native.load()
    before = _native.route_count()
    pyr = [[45, 80], [23, 40], [12, 20], [6, 10]]
    kf = _native.route_key(False, 0, 16, 6, 5, 4820, 8, 32, 4, 300, 4, 4, pyr)
    kb = _native.route_key(True, 0, 16, 6, 5, 4820, 8, 32, 4, 300, 4, 4, torch.tensor(pyr))
    assert kf == "f|0|16|6|5|4820|8|32|4|300|4|4|45x80,23x40,12x20,6x10" and kb == "b" + kf[1:]
    assert _native.route_key(False, 2, 8, 1, 0, 22223, 8, 32, 4, 22223, 4, 4, [[100, 167], [50, 84], [25, 42], [13, 21]]) != kf
    try:
        _native.pin_route(kf, {"fwd_rs": 1, "fwd_rs_nt": 2})
        _native.pin_route(kf, "fwd_win=1")                         # replaces
        _native.pin_route(kb, {"bwd_rs_fsplit": 4, "scatter_order": 1, "scatter_mfma": 0})
        assert _native.route_count() == before + 2
        with pytest.raises(RuntimeError, match="cannot parse"):
            _native.pin_route(kb, "tiles=3")
        _native.pin_route(kf, "")                                  # removes
        assert _native.route_count() == before + 1
        path = tmp_path / "routes.json"
        path.write_text(json.dumps({"device": "test", "routes": {kf: {"fwd_rs_nt": 1}, kb: {"bwd_win": 0}}}))
        assert _native.load_routes(str(path)) == 2 and _native.route_count() == before + 2
    finally:
        _native.pin_route(kf, "")
        _native.pin_route(kb, "")
    assert _native.route_count() == before
    shipped = _native.ROUTES_FILE
    if os.path.exists(shipped):                                    # the audited table parses and every entry is accepted
        doc = json.load(open(shipped))
        assert set(doc) >= {"device", "routes"} and all(k[0] in "fb" for k in doc["routes"])
        assert _native.route_count() >= min(1, len(doc["routes"]))


def test_graphed_layer_signatures_and_cpu_refusal():
    """devis_amd.graphed keys its captured graphs by the shapes / dtypes / requires_grad of the floating-point arguments and by the
    IDENTITY (object + version) of everything else -- the integer tensors whose host copy chooses the kernels; CPU tensors are
    refused (there is no CPU path to capture)."""
    import devis_amd
    from devis_amd.graphs import GraphedLayer
    layer = GraphedLayer(torch.nn.Identity())
    shapes, offs = torch.tensor([[4, 5], [2, 3]]), [torch.tensor([1]), torch.tensor([-1])]
    q = torch.zeros(1, 12, 8)
    sig = layer._signature((q, shapes, (shapes, shapes), offs, None, 64))
    assert sig == layer._signature((torch.ones(1, 12, 8), shapes, (shapes, shapes), list(offs), None, 64))      # values of flowing tensors: no
    assert sig != layer._signature((torch.zeros(1, 13, 8), shapes, (shapes, shapes), offs, None, 64))          # shape: yes
    assert sig != layer._signature((q.double(), shapes, (shapes, shapes), offs, None, 64))                     # dtype: yes
    assert sig != layer._signature((q.clone().requires_grad_(True), shapes, (shapes, shapes), offs, None, 64))
    assert sig != layer._signature((q, shapes.clone(), (shapes, shapes), offs, None, 64))                      # another shapes OBJECT: yes
    assert sig != layer._signature((q, shapes, (shapes, shapes), offs, None, 2))                               # plain values: by value
    shapes.add_(0)
    assert sig != layer._signature((q, shapes, (shapes, shapes), offs, None, 64))                              # in-place change: version
    with pytest.raises(RuntimeError, match="no CPU path"):
        devis_amd.graphed(torch.nn.Identity(), (q,))


def test_frame_tables_of_two_stacks_do_not_evict_each_other_and_are_thread_safe():
    """The encoder and the decoder stack hand different offset lists to their layers in turn (devis_transformer.py:103-121,
    151-169): each list keeps its own entry (round 4 had ONE class-level slot that the two stacks thrashed), and concurrent
    forwards from several threads never see a half-updated cache."""
    import threading
    from devis_amd.modules import TemporalMSDeformAttnDecoder as Dec, TemporalMSDeformAttnEncoder as Enc
    enc = [torch.tensor([1, 2]), torch.tensor([-1, 1]), torch.tensor([-2, -1])]
    dec = [torch.tensor([2, 1]), torch.tensor([1, -1]), torch.tensor([-1, -2])]
    a, b = Enc._frame_table(enc, 3, torch.device("cpu")), Dec._frame_table(dec, 3, torch.device("cpu"))
    for _ in range(3):                      # layer after layer, alternating stacks: always hits
        assert Enc._frame_table(enc, 3, torch.device("cpu")) is a
        assert Dec._frame_table(dec, 3, torch.device("cpu")) is b
    errors = []

    def worker(seed):
        try:
            g = torch.Generator().manual_seed(seed)
            for _ in range(200):
                offs = [torch.randint(-f, 3 - f, (2,), generator=g) for f in range(3)]
                want = [[int(o[0]) + f, int(o[1]) + f] for f, o in enumerate(offs)]
                got = Dec._frame_table(offs, 3, torch.device("cpu")).tolist()
                assert got == [[v % 3 for v in row] for row in want]
                assert Enc._frame_table(enc, 3, torch.device("cpu")).tolist() == a.tolist()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_frame_tables_served_inside_a_graph_capture_are_never_freed(monkeypatch):
    """A HIP graph holds the ADDRESS of the frame table it was captured with; the cache is a small LRU, so a table served
    while the stream is capturing is kept for good (here the capture is simulated: no GPU)."""
    import weakref
    from devis_amd.modules import ms_deform_attn as mm
    cache = mm._FrameTables()
    first = [torch.tensor([1, 2]), torch.tensor([-1, 1]), torch.tensor([-2, -1])]
    eager = cache.get(first, 3, torch.device("cpu"))            # built and cached outside a capture ...
    monkeypatch.setattr(mm, "_capturing", lambda t: True)
    ref = weakref.ref(cache.get(first, 3, torch.device("cpu")))  # ... and served inside one
    assert ref() is eager
    del eager
    other = [torch.tensor([2, 1]), torch.tensor([1, -1]), torch.tensor([-1, -2])]
    built_inside = cache.get(other, 3, torch.device("cpu"))      # a table BUILT inside a capture is that graph's temporary: not cached
    assert not any(cache._same(e[0], other) for e in cache._entries) and built_inside.tolist() == [[2, 1], [2, 0], [1, 0]]
    monkeypatch.setattr(mm, "_capturing", lambda t: False)
    for i in range(cache.capacity + 3):                 # push the first entry out of the LRU
        cache.get([torch.tensor([1, 2]), torch.tensor([-1, 1]), torch.tensor([-2, -1])], 3, torch.device("cpu"))
    assert not any(cache._same(e[0], first) for e in cache._entries)
    assert ref() is not None and ref().tolist() == [[1, 2], [0, 2], [0, 1]]


def test_bad_cpu_offsets_raise_with_the_offsets_in_the_message():
    from devis_amd.modules import TemporalMSDeformAttnDecoder as Dec
    with pytest.raises(IndexError, match=r"outside the clip's 3 frames \(temporal_offsets = \[\[1, 7\]"):
        Dec._frame_table([torch.tensor([1, 7])] * 3, 3, torch.device("cpu"))


def test_project_value_padded_equals_dense_linear(monkeypatch):
    """functions.project_value (SURVEY f-3): value_proj written with a padded pixel stride and masked by
    msda_mask_rows (here: its CPU test double) gives the same value tensor and the same gradients (input, weight,
    bias) as the reference's dense Linear + masked_fill (ms_deform_attn.py:101-103); only the strides differ.  The
    caller's cotangent is never modified."""
    from devis_amd.functions import project_value
    fake_native.install(monkeypatch)
    torch.manual_seed(3)
    lin = torch.nn.Linear(24, 32).double()
    x = torch.randn(2, 13, 24, dtype=torch.float64, requires_grad=True)
    mask = torch.rand(2, 13) < 0.3
    cot = torch.randn(2, 13, 4, 8, dtype=torch.float64)
    cot0 = cot.clone()
    ref = lin(x).masked_fill(mask[..., None], float(0)).view(2, 13, 4, 8)           # the reference's lines
    res = [[ref.detach().clone()] + [t.clone() for t in torch.autograd.grad(ref, (x, lin.weight, lin.bias), cot)]]
    for pad in (0, 1, 3):
        v = project_value(x, lin, 4, mask, pad_heads=pad)
        assert v.shape == (2, 13, 4, 8)
        assert v.is_contiguous() == (pad == 0)
        if pad:
            assert v.stride() == (13 * (4 + pad) * 8, (4 + pad) * 8, 8, 1)
        g = torch.autograd.grad(v, (x, lin.weight, lin.bias), cot)
        res.append([v.detach().clone()] + [t.clone() for t in g])
    # consumer_masks_grad: the backward trusts the incoming gradient to be masked already
    v = project_value(x, lin, 4, mask, pad_heads=1, consumer_masks_grad=True)
    g = torch.autograd.grad(v, (x, lin.weight, lin.bias), cot.masked_fill(mask[..., None, None], 0.0))
    res.append([v.detach().clone()] + [t.clone() for t in g])
    for other in res[1:]:
        for a, b in zip(res[0], other):
            torch.testing.assert_close(a, b, rtol=1e-12, atol=1e-13)
    assert (res[1][0][mask] == 0).all()
    assert torch.equal(cot, cot0)
    assert project_value(x, lin, 4, None, pad_heads=0).is_contiguous()


def test_cached_reference_points_match_the_reference_call_site():
    """the cached get_reference_points equals, bit for bit, what the REFERENCE's
    DeformableTransformerEncoder.get_reference_points (deformable_transformer.py:185-198) returned for the same pyramid
    and valid ratios (tests/golden/args_call_sites.npz, recorded from the reference's encoder stack), and
    patch_transformer() installs it on a class shaped like the reference's."""
    import types
    import devis_amd
    from devis_amd import argument_builders as ab
    g = golden("args_call_sites")
    ss = torch.from_numpy(g["spatial_shapes"])
    for T in (5, 6):
        vr = torch.from_numpy(g["T%d/valid_ratios" % T])
        want = torch.from_numpy(g["T%d/reference_points" % T])
        assert torch.equal(ab.get_reference_points(ss, vr, torch.device("cpu")), want)
        assert torch.equal(ab.get_reference_points([tuple(r) for r in ss.tolist()], vr, torch.device("cpu")), want)   # cache hit
    assert len(ab._grid_cache) == 1

    class DeformableTransformerEncoder:                      # the attribute layout of deformable_transformer.py:178-198
        @staticmethod
        def get_reference_points(spatial_shapes, valid_ratios, device):
            raise AssertionError("the un-patched builder ran")

        def forward(self, spatial_shapes, valid_ratios):
            return self.get_reference_points(spatial_shapes, valid_ratios, device=valid_ratios.device)

    class DeVISTransformerEncoder(DeformableTransformerEncoder):       # devis_transformer.py:83
        pass

    mod = types.SimpleNamespace(DeformableTransformerEncoder=DeformableTransformerEncoder)
    previous = devis_amd.patch_transformer(mod)
    assert isinstance(previous, staticmethod)
    assert torch.equal(DeVISTransformerEncoder().forward(ss, vr), want)


@pytest.mark.parametrize("mode", ["enc_all", "enc_win2", "enc_win4", "dec"])
@pytest.mark.parametrize("T", [5, 6])
def test_frame_table_selects_the_frames_the_reference_indexes(T, mode):
    """from the temporal_offsets the REFERENCE's stacks build (devis_transformer.py:97-118, 146-153;
    recorded in args_call_sites.npz -- connect-all, windows of 2 and 4 with mirrored clip ends) _frame_table derives
    exactly the frames value[temporal_offsets[t] + t] (ms_deform_attn.py:339,445) selects."""
    from devis_amd.modules.ms_deform_attn import TemporalMSDeformAttnBase
    g = golden("args_call_sites")
    offs = [torch.from_numpy(r.copy()) for r in g["T%d/%s/offsets" % (T, mode)]]
    table = TemporalMSDeformAttnBase._frame_table(offs, T, torch.device("cpu"))
    assert table.dtype == torch.int32 and table.tolist() == g["T%d/%s/frames" % (T, mode)].tolist()


def test_frame_table_rejects_out_of_range_offsets():
    """A negative index wraps once, anything else out of range is an error: IndexError for host tensors;
    for device tensors the same asynchronous device-side assert the reference's own indexing kernel raises."""
    from devis_amd.modules.ms_deform_attn import TemporalMSDeformAttnBase
    T = 4
    for bad in ([torch.tensor([1, 2]) for _ in range(T)],                  # frame T-1 + 1 = T
                [torch.tensor([-1, -2 * T]) for _ in range(T)]):           # wraps more than once
        with pytest.raises(IndexError):
            TemporalMSDeformAttnBase._frame_table(bad, T, torch.device("cpu"))


def test_fused_linear_parameters_are_cached_only_outside_autograd():
    """the concatenated query-side Linear parameters are rebuilt per call while autograd records
    (their gradients must reach the individual Linears) and cached per parameter version otherwise."""
    from devis_amd.modules import MSDeformAttn
    from devis_amd.modules.ms_deform_attn import _fused_linear_params
    mod = MSDeformAttn(32, 2, 4, 3)
    lins = (mod.sampling_offsets, mod.attention_weights)
    w1, b1 = _fused_linear_params(mod, lins)
    assert w1.requires_grad and w1.shape == (4 * 2 * 3 * 3, 32)
    with torch.no_grad():
        w2, b2 = _fused_linear_params(mod, lins)
        w3, b3 = _fused_linear_params(mod, lins)
        assert w2 is w3 and b2 is b3 and not w2.requires_grad and torch.equal(w2, w1.detach())
        mod.attention_weights.bias.add_(1.0)                     # a parameter changes in place: new version -> rebuilt
        w4, b4 = _fused_linear_params(mod, lins)
        assert b4 is not b2 and torch.equal(b4[-mod.attention_weights.bias.numel():], mod.attention_weights.bias)


def test_split_k_weight_gradient_of_value_proj_matches_the_plain_product():
    """`_split_k_wgrad` (the weight gradient of the padded value_proj, cut into row slices) against g.t() @ x, with a row
    count that leaves a tail slice, in fp64; and through the autograd Function against nn.Linear's own backward."""
    import torch
    from devis_amd.functions import project_value
    from devis_amd.functions.ms_deform_attn_func import _split_k_wgrad
    gen = torch.Generator().manual_seed(5)
    g = torch.randn(4 * 1024 + 37, 24, generator=gen, dtype=torch.float64)
    x = torch.randn(4 * 1024 + 37, 16, generator=gen, dtype=torch.float64)
    torch.testing.assert_close(_split_k_wgrad(g, x), g.t() @ x, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(_split_k_wgrad(g[:100], x[:100]), g[:100].t() @ x[:100], rtol=0, atol=0)      # short: plain product
    lin = torch.nn.Linear(16, 24).double()
    inp = torch.randn(3, 1500, 16, generator=gen, dtype=torch.float64, requires_grad=True)
    wgt = torch.randn(3, 1500, 4, 6, generator=gen, dtype=torch.float64)
    got = torch.autograd.grad((project_value(inp, lin, 4, None, 1) * wgt).sum(), (inp, lin.weight, lin.bias))
    want = torch.autograd.grad((lin(inp).view(3, 1500, 4, 6) * wgt).sum(), (inp, lin.weight, lin.bias))
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=1e-11, atol=1e-11)


def test_split_k_weight_gradient_on_16_bit_inputs_is_as_accurate_as_one_gemm():
    """16-bit partial products must not be rounded before they are summed.  On the CPU torch has no fp32-output bmm,
    so `_split_k_wgrad` takes the single GEMM (identical error); the fp32-partials branch is checked with a stand-in for
    `_bmm_f32` (float casts -- what the GPU kernel computes), and on the GPU itself in tests/test_modules_gpu.py."""
    from devis_amd.functions import ms_deform_attn_func as F
    gen = torch.Generator().manual_seed(11)
    R = 28 * 1024 + 200
    g = torch.randn(R, 32, generator=gen).to(torch.bfloat16)
    x = torch.randn(R, 16, generator=gen).to(torch.bfloat16)
    exact = g.double().t() @ x.double()
    rms = lambda w: float(((w.double() - exact) ** 2).mean().sqrt() / (exact ** 2).mean().sqrt())
    single = rms(g.t() @ x)
    assert rms(F._split_k_wgrad(g, x)) <= single * 1.001          # CPU: falls back to the single GEMM
    old = F._bmm_f32
    F._bmm_f32 = lambda a, b: torch.bmm(a.float(), b.float())
    try:
        split = rms(F._split_k_wgrad(g, x))
    finally:
        F._bmm_f32 = old
    assert split <= single * 1.05, (split, single)
    rounded = rms((torch.bmm(g[:28 * 1024].view(28, 1024, -1).transpose(1, 2), x[:28 * 1024].view(28, 1024, -1)).sum(0, dtype=torch.float32)
                   + (g[28 * 1024:].t() @ x[28 * 1024:]).float()).to(torch.bfloat16))
    assert rounded > single * 1.2                                   # what round 5 shipped: measurably worse
