"""The two integration paths INTEGRATION.md documents, executed (SURVEY section 8, row b).

Path A: the package reached as ``src.models.ops`` through a symlink, the way DeVIS imports it
(/root/reference/src/models/deformable_transformer.py:17, devis_transformer.py:13).
Path B: ``integration/MultiScaleDeformableAttention.py`` -- the module DeVIS's own functions file imports -- put on the import
path and driven by an autograd.Function shaped like the reference's binding
(/root/reference/src/models/ops/functions/ms_deform_attn_func.py:18-38); also as ``pip`` installs it (pyproject.toml /
setup.py, the counterpart of the reference's src/models/ops/setup.py:36-71 + make.sh).

The CPU tests check that both import and load the library (no compute); the GPU tests run golden fixtures through them.
"""
import importlib
import os
import subprocess
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


@pytest.fixture()
def msda_stub(tmp_path):
    """integration/MultiScaleDeformableAttention.py copied to <tmp> (as a maintainer would drop it next to main.py) and imported
    under the name DeVIS imports."""
    import shutil
    from devis_amd import build
    build.ensure()
    shutil.copy(os.path.join(ROOT, "integration", "MultiScaleDeformableAttention.py"), tmp_path / "MultiScaleDeformableAttention.py")
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


def test_pip_install_builds_and_ships_the_library_and_the_reference_module_name(tmp_path):
    """`pip install .` (offline: --no-build-isolation --no-deps) into a scratch target: the wheel holds the devis_amd package
    WITH libmsda_hip.so, the kernel sources and the header, and the top-level module the reference imports; a fresh
    interpreter that sees only the scratch target imports both and loads the installed library (no compute)."""
    import glob
    import shutil
    target = tmp_path / "site"
    litter = [q for q in [os.path.join(ROOT, "build")] + glob.glob(os.path.join(ROOT, "*.egg-info")) if not os.path.exists(q)]
    r = subprocess.run([sys.executable, "-m", "pip", "install", "--no-build-isolation", "--no-deps", "--no-index", "--quiet",
                        "--target", str(target), ROOT], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    for q in litter + glob.glob(os.path.join(ROOT, "*.egg-info")):       # what setuptools' in-tree build leaves behind
        if q in litter or q.endswith(".egg-info"):
            shutil.rmtree(q, ignore_errors=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for rel in ("MultiScaleDeformableAttention.py", "devis_amd/libmsda_hip.so", "devis_amd/include/msda.h",
                "devis_amd/csrc/msda_api.hip", "devis_amd/modules/ms_deform_attn.py"):
        assert (target / rel).exists(), rel
    probe = ("import sys, os; sys.path.insert(0, %r); import devis_amd, devis_amd._native as n, devis_amd.build as b; "
             "assert os.path.dirname(b.lib_path()) == os.path.join(%r, 'devis_amd'), b.lib_path(); "
             "assert not b.is_stale(); lib = n.load(); assert lib.msda_version() == n.MSDA_ABI_VERSION; "
             "import MultiScaleDeformableAttention as M; "
             "assert M.__file__.startswith(%r) and callable(M.ms_deform_attn_forward) and callable(M.ms_deform_attn_backward); "
             "print('installed ok')" % (str(target), str(target), str(target)))
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "MSDA_LIB")}
    r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=env)
    assert r.returncode == 0 and "installed ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


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
