"""The two integration paths INTEGRATION.md documents, executed (SURVEY section 8, row b).

Path A: the package reached as ``src.models.ops`` through a symlink, the way DeVIS imports it
(/root/reference/src/models/deformable_transformer.py:17, devis_transformer.py:13).
Path B: the ``MultiScaleDeformableAttention.py`` ctypes stub printed in INTEGRATION.md, extracted from the document
as is, driven by an autograd.Function shaped like the reference's binding
(/root/reference/src/models/ops/functions/ms_deform_attn_func.py:18-38).

The CPU tests check that both import and load the library (no compute); the GPU tests run golden fixtures through them.
"""
import importlib
import os
import re
import sys

import numpy as np
import pytest
import torch

import module_cases
from conftest import ROOT, golden


@pytest.fixture()
def ops_as_reference_package(tmp_path):
    """<tmp>/src/models/ops -> devis_amd (symlink), <tmp> on sys.path: yields the package imported as src.models.ops."""
    models = tmp_path / "src" / "models"
    models.mkdir(parents=True)
    os.symlink(os.path.join(ROOT, "devis_amd"), models / "ops")
    sys.path.insert(0, str(tmp_path))
    before = set(sys.modules)
    try:
        yield importlib.import_module("src.models.ops")
    finally:
        sys.path.remove(str(tmp_path))
        for name in set(sys.modules) - before:
            if name == "src" or name.startswith("src."):
                del sys.modules[name]


def _stub_source():
    """The MultiScaleDeformableAttention.py block of INTEGRATION.md, with the library path filled in."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if b.lstrip().startswith("# MultiScaleDeformableAttention.py")]
    assert len(stub) == 1, "INTEGRATION.md must hold exactly one MultiScaleDeformableAttention.py block"
    from devis_amd import build
    src = stub[0].replace("/path/to/devis_amd/libmsda_hip.so", build.lib_path())
    assert build.lib_path() in src
    return src


@pytest.fixture()
def msda_stub(tmp_path):
    """The stub written to <tmp>/MultiScaleDeformableAttention.py and imported under the name DeVIS imports."""
    from devis_amd import build
    build.ensure()
    (tmp_path / "MultiScaleDeformableAttention.py").write_text(_stub_source())
    sys.path.insert(0, str(tmp_path))
    try:
        sys.modules.pop("MultiScaleDeformableAttention", None)
        yield importlib.import_module("MultiScaleDeformableAttention")
    finally:
        sys.path.remove(str(tmp_path))
        sys.modules.pop("MultiScaleDeformableAttention", None)


def _reference_shaped_function(MSDA):
    """An autograd.Function with the contract of the reference's MSDeformAttnFunction (ms_deform_attn_func.py:21-38)
    over the module `MSDA` -- what DeVIS's own, unmodified functions file is once `import
    MultiScaleDeformableAttention as MSDA` resolves to the stub."""
    from torch.autograd import Function
    from torch.autograd.function import once_differentiable

    class RefShaped(Function):
        @staticmethod
        def forward(ctx, value, shapes, lsi, loc, aw, im2col_step):
            ctx.im2col_step = im2col_step
            out = MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, ctx.im2col_step)
            ctx.save_for_backward(value, shapes, lsi, loc, aw)
            return out

        @staticmethod
        @once_differentiable
        def backward(ctx, grad_output):
            value, shapes, lsi, loc, aw = ctx.saved_tensors
            gv, gl, ga = MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, aw, grad_output, ctx.im2col_step)
            return gv, None, None, gl, ga, None

    return RefShaped


# ---- CPU: both paths import and load (no compute) -------------------------------------------------------------------

def test_path_a_package_imports_and_loads_under_the_reference_name(ops_as_reference_package):
    ops = ops_as_reference_package
    mods = importlib.import_module("src.models.ops.modules")
    fns = importlib.import_module("src.models.ops.functions")
    for name in ("MSDeformAttn", "TemporalMSDeformAttnEncoder", "TemporalMSDeformAttnDecoder"):      # deformable_transformer.py:17
        assert hasattr(mods, name)
    assert hasattr(fns, "MSDeformAttnFunction") and hasattr(fns, "ms_deform_attn_core_pytorch")
    native = importlib.import_module("src.models.ops._native")
    assert native is not importlib.import_module("devis_amd._native")          # a second, independent import
    build = importlib.import_module("src.models.ops.build")
    assert build.is_stale() is False                                           # (raised FileNotFoundError here)
    assert os.path.samefile(build.lib_path(), os.path.join(ROOT, "devis_amd", "libmsda_hip.so"))
    lib = native.load()
    assert lib.msda_version() == native.MSDA_ABI_VERSION
    assert ops.__name__ == "src.models.ops"


def test_path_b_stub_imports_and_binds_the_library(msda_stub):
    assert callable(msda_stub.ms_deform_attn_forward) and callable(msda_stub.ms_deform_attn_backward)    # vision.cpp:14-15
    g = golden("op_testpy_shape")
    v = torch.from_numpy(g["value"])
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):          # ms_deform_attn.h:38
        msda_stub.ms_deform_attn_forward(v, torch.from_numpy(g["spatial_shapes"]), torch.from_numpy(g["level_start_index"]),
                                         torch.from_numpy(g["sampling_locations"]), torch.from_numpy(g["attention_weights"]), 2)


# ---- GPU: golden fixtures through both paths ------------------------------------------------------------------------

def _op_case(name, dtype, dev):
    g = golden(name)
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    v, l, a = (t(k).to(dtype).requires_grad_(True) for k in ("value", "sampling_locations", "attention_weights"))
    return g, v, t("spatial_shapes"), t("level_start_index"), l, a, t("grad_output").to(dtype)


def _check_op(g, out, grads, rtol, atol):
    scale = lambda k: max(1.0, float(np.abs(g[k]).max()))
    np.testing.assert_allclose(out.detach().double().cpu().numpy(), g["out"], rtol=rtol, atol=atol * scale("out"))
    for got, key in zip(grads, ("grad_value", "grad_sampling_loc", "grad_attn_weight")):
        ref, have = g[key], got.double().cpu().numpy()
        if key == "grad_sampling_loc" and "loc_grad_mask" in g:       # exact-border points (tests/golden/make_golden.py)
            ref, have = ref * g["loc_grad_mask"], have * g["loc_grad_mask"]
        np.testing.assert_allclose(have, ref, rtol=rtol, atol=atol * scale(key))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,rtol,atol", [(torch.float64, 1e-9, 1e-11), (torch.float32, 2e-4, 2e-5)], ids=["f64", "f32"])
def test_path_a_operator_and_module_through_src_models_ops(ops_as_reference_package, dtype, rtol, atol):
    fns = importlib.import_module("src.models.ops.functions")
    g, v, ss, lsi, l, a, go = _op_case("op_devis_small", dtype, "cuda")
    out = fns.MSDeformAttnFunction.apply(v, ss, lsi, l, a, 64)
    _check_op(g, out, torch.autograd.grad(out, (v, l, a), go), rtol, atol)
    native = importlib.import_module("src.models.ops._native")
    assert "msda" in native.last_route()                                        # the HIP library ran, through THIS import
    mods = importlib.import_module("src.models.ops.modules")
    got, gm = module_cases.run("mod_temporal_dec_ref2", "cuda", dtype, fused=True, modules=mods)
    module_cases.compare(got, gm, rtol=max(rtol, 1e-9) * (1 if dtype == torch.float64 else 5), atol=atol * 10)


@pytest.mark.gpu
@pytest.mark.parametrize("step", [2, 64])
@pytest.mark.parametrize("dtype,rtol,atol", [(torch.float64, 1e-9, 1e-11), (torch.float32, 2e-4, 2e-5)], ids=["f64", "f32"])
def test_path_b_stub_under_a_reference_shaped_function(msda_stub, dtype, rtol, atol, step):
    fn = _reference_shaped_function(msda_stub)
    for name in ("op_batched_im2col", "op_devis_small"):
        g, v, ss, lsi, l, a, go = _op_case(name, dtype, "cuda")
        if v.shape[0] % min(v.shape[0], step):
            continue
        out = fn.apply(v, ss, lsi, l, a, step)
        _check_op(g, out, torch.autograd.grad(out, (v, l, a), go), rtol, atol)
    g, v, ss, lsi, l, a, go = _op_case("op_batched_im2col", dtype, "cuda")
    with pytest.raises(RuntimeError, match="must divide"):                      # ms_deform_attn_cuda.cu:52
        fn.apply(v, ss, lsi, l, a, 4)
