"""End-to-end drop-in check against the reference's OWN transformer stack (SURVEY section 8, row b) --
build container only: it needs /root/reference and is skipped everywhere else (nothing of the reference travels).

A scratch package tree is assembled from SYMLINKS to the reference's unmodified `deformable_transformer.py` and
`devis_transformer.py` (/root/reference/src/models/devis_transformer.py:16-88, deformable_transformer.py:17,52-55), with
`src/models/ops` pointing once at `devis_amd` and once at the reference's `ops`; the only stand-ins are an empty
`src/models/__init__.py` (the reference's imports backbones and torchvision), `src/util/misc.inverse_sigmoid` (misc.py
imports visdom / torchvision, absent here) and -- for the reference's `ops` -- a `MultiScaleDeformableAttention` module whose
two entry points run the CPU oracle (the compiled CUDA extension does not exist here).  `devis_amd`'s kernels are replaced by
the oracle-backed double of tests/fake_native.py, so the whole stack runs on CPU tensors.

Checked: the DeVISTransformer built on either `ops` has the same state-dict keys, shapes and -- with the same seed -- bit-identical
initial values; a state dict moves between them; `devis_amd.patch_transformer` gives bit-identical encoder reference points on
the real class; one encoder layer + one decoder layer forward agree.
"""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

import fake_native
from conftest import ROOT
from oracle import msda_oracle as O

REF_SRC = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF_SRC, "models", "devis_transformer.py")),
                                reason="needs the reference tree (build container only)")


def _inverse_sigmoid_source():
    # our own statement of the helper the transformer imports from util.misc (logit with clamping)
    return ("import torch\n\n\n"
            "def inverse_sigmoid(x, eps=1e-5):\n"
            "    x = x.clamp(min=0, max=1)\n"
            "    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))\n")


def _tree(root, ops_target):
    models, util = root / "src" / "models", root / "src" / "util"
    models.mkdir(parents=True)
    util.mkdir()
    for d in (root / "src", models, util):
        (d / "__init__.py").write_text("")
    (util / "misc.py").write_text(_inverse_sigmoid_source())
    for name in ("deformable_transformer.py", "devis_transformer.py"):
        os.symlink(os.path.join(REF_SRC, "models", name), models / name)       # the reference's files, unmodified, not copied
    os.symlink(ops_target, models / "ops")


def _oracle_extension():
    """`MultiScaleDeformableAttention` (vision.cpp:13-16) on CPU tensors through the oracle."""
    mod = types.ModuleType("MultiScaleDeformableAttention")
    npf = lambda t: t.detach().double().cpu().numpy()

    def ms_deform_attn_forward(value, shapes, lsi, loc, aw, im2col_step):
        return torch.from_numpy(O.forward(npf(value), shapes.numpy(), lsi.numpy(), npf(loc), npf(aw))).to(value.dtype)

    def ms_deform_attn_backward(value, shapes, lsi, loc, aw, grad_output, im2col_step):
        gv, gl, ga = O.backward(npf(value), shapes.numpy(), lsi.numpy(), npf(loc), npf(aw), npf(grad_output))
        return [torch.from_numpy(x).to(value.dtype) for x in (gv, gl, ga)]

    mod.ms_deform_attn_forward, mod.ms_deform_attn_backward = ms_deform_attn_forward, ms_deform_attn_backward
    return mod


class _Imported:
    """The scratch tree on sys.path; every `src*` module (and the extension stand-in) is dropped again on exit."""

    def __init__(self, root, extension=None):
        self.root, self.extension = str(root), extension

    def __enter__(self):
        self.before = set(sys.modules)
        sys.path.insert(0, self.root)
        if self.extension is not None:
            sys.modules["MultiScaleDeformableAttention"] = self.extension
        return importlib.import_module("src.models.devis_transformer")

    def __exit__(self, *exc):
        sys.path.remove(self.root)
        for name in set(sys.modules) - self.before:
            if name == "src" or name.startswith("src.") or name == "MultiScaleDeformableAttention":
                del sys.modules[name]
        return False


CFG = dict(d_model=64, num_frames=3, nhead=8, num_encoder_layers=1, num_decoder_layers=1, dim_feedforward=96, dropout=0.0,
           num_feature_levels=2, enc_connect_all_embeddings=True, enc_n_curr_points=4, enc_n_temporal_points=2,
           dec_n_curr_points=4, dec_n_temporal_points=2, dec_instance_aware_att=True)
PYRAMID = [(6, 8), (3, 4)]


def _inputs():
    g = torch.Generator().manual_seed(3)
    T, C = CFG["num_frames"], CFG["d_model"]
    srcs = [torch.randn(T, C, h, w, generator=g) for h, w in PYRAMID]
    masks = [torch.zeros(T, h, w, dtype=torch.bool) for h, w in PYRAMID]
    for m in masks:
        m[:, :, -1] = True          # one padded column: valid ratios < 1 and a padding mask for value_proj
    pos = [torch.randn(T, C, h, w, generator=g) * 0.1 for h, w in PYRAMID]
    query_embed = torch.randn(2 * T, 2 * C, generator=g)      # 2 queries per frame (the decoder reshapes them frame-major)
    return srcs, masks, pos, query_embed


def _build_and_run(module):
    torch.manual_seed(1234)
    model = module.DeVISTransformer(**CFG).eval()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        hs, _, memories, init_ref, inter_ref, lsi, valid_ratios, shapes = model(*_inputs())
    return model, state, dict(hs=hs, memory0=memories[0], memory1=memories[1], init_ref=init_ref, inter_ref=inter_ref,
                              valid_ratios=valid_ratios)


def test_devis_transformer_on_devis_amd_equals_the_one_on_the_reference_ops(tmp_path, monkeypatch):
    a, b = tmp_path / "on_devis_amd", tmp_path / "on_reference_ops"
    _tree(a, os.path.join(ROOT, "devis_amd"))
    _tree(b, os.path.join(REF_SRC, "models", "ops"))

    with _Imported(a) as mod_a:
        native = importlib.import_module("src.models.ops._native")
        fake_native.install(monkeypatch, native=native,
                            functions=importlib.import_module("src.models.ops.functions.ms_deform_attn_func"))
        assert os.path.samefile(os.path.dirname(native.__file__), os.path.join(ROOT, "devis_amd"))
        model_a, state_a, out_a = _build_and_run(mod_a)
        # the cached argument builder on the REAL encoder class: bit-identical reference points
        dt = importlib.import_module("src.models.deformable_transformer")
        ops_pkg = importlib.import_module("src.models.ops")
        shapes = torch.tensor(PYRAMID, dtype=torch.long)
        vr = out_a["valid_ratios"]
        want = dt.DeformableTransformerEncoder.get_reference_points(shapes, vr, device=vr.device)
        srcs, masks, pos, _ = _inputs()
        with torch.no_grad():
            prepared = model_a.prepare_data(srcs, masks, pos)          # the reference's own method (deformable_transformer.py:69-94)
        previous = ops_pkg.patch_transformer(dt, mod_a)
        try:
            got = dt.DeformableTransformerEncoder.get_reference_points(shapes, vr, device=vr.device)
            assert torch.equal(got, want)
            # the replacement prepare_data on the REAL class -- the same six results, bit for bit, with the two
            # pyramid tensors interned (the same objects on every call, their host values known to the binding)
            with torch.no_grad():
                mine, mine2 = model_a.prepare_data(srcs, masks, pos), model_a.prepare_data(srcs, masks, pos)
            assert len(mine) == len(prepared) == 6
            for x, y in zip(mine, prepared):
                assert x.dtype == y.dtype and torch.equal(x, y)
            assert mine[3] is mine2[3] and mine[4] is mine2[4] and mine[3] is not prepared[3]
            assert native.known_host_values(mine[3]) == tuple(v for hw in PYRAMID for v in hw)
            # ... and the stacks' temporal offsets (devis_transformer.py:100, 149) come out interned through the module's `torch`
            a1 = mod_a.torch.tensor([1, 2], device=vr.device)
            assert a1 is mod_a.torch.tensor([1, 2], device=vr.device) and a1.tolist() == [1, 2]
            assert mod_a.torch.tensor([1.5], device=vr.device).dtype == torch.float32 and mod_a.torch.cat is torch.cat
            with torch.no_grad():
                again = model_a(*_inputs())[0]
            assert torch.equal(again, out_a["hs"])                 # the patched stack computes the same thing
        finally:
            ops_pkg.argument_builders.unpatch_transformer(dt, previous, mod_a)
        assert mod_a.torch is torch and dt.DeformableTransformer.prepare_data.__qualname__ == "DeformableTransformer.prepare_data"
        assert type(model_a.encoder.layers[0].self_attn).__module__.startswith("src.models.ops.modules")

    with _Imported(b, extension=_oracle_extension()) as mod_b:
        model_b, state_b, out_b = _build_and_run(mod_b)
        ref_attn = type(model_b.decoder.layers[0].cross_attn)
        assert ref_attn.__name__ == "TemporalMSDeformAttnDecoder"
        assert os.path.samefile(sys.modules[ref_attn.__module__].__file__,
                                os.path.join(REF_SRC, "models", "ops", "modules", "ms_deform_attn.py"))
        # a state dict of the devis_amd-backed model loads into the reference-backed one
        model_b.load_state_dict(state_a, strict=True)

    assert list(state_a) == list(state_b) and len(state_a) >= 40
    for k in state_a:
        assert state_a[k].shape == state_b[k].shape, k
        assert torch.equal(state_a[k], state_b[k]), "same-seed initialisation differs at %s" % k
    for k in out_a:
        x, y = out_a[k].double().numpy(), out_b[k].double().numpy()
        assert x.shape == y.shape, k
        assert np.abs(x - y).max() <= 2e-5 * max(1.0, np.abs(y).max()), (k, np.abs(x - y).max())
