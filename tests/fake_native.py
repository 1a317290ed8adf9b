"""Test double for devis_amd._native (tests/ only): runs the C-ABI entry points' CONTRACT on CPU tensors
through the CPU oracle, so the host logic (autograd wiring, chunking, module arithmetic) can be tested
without a GPU.  The product never imports this."""
import numpy as np
import torch

from helpers import temporal_reference
from oracle import msda_oracle as O


def _np(t):
    return t.detach().double().cpu().numpy()


def install(monkeypatch, native=None, functions=None):
    """Patch devis_amd so CPU tensors are accepted and routed to the oracle.  `native` / `functions`: the `_native` and
    `functions.ms_deform_attn_func` module objects to patch when the package was imported under another name (as
    `src.models.ops`: tests/test_reference_stack_cpu.py)."""
    if native is None:
        from devis_amd import _native
    else:
        _native = native
    if functions is None:
        from devis_amd.functions import ms_deform_attn_func as F
    else:
        F = functions

    def _check_inputs(named):
        for name, t in named:
            if name == "value" and t.dim() == 4:
                _native.value_strides(t)             # dense, head-major or padded rows
            elif not t.is_contiguous():
                raise RuntimeError("%s tensor has to be contiguous" % name)

    def forward(value, shapes, lsi, loc, aw, out):
        r = O.forward(_np(value), shapes.numpy(), lsi.numpy(), _np(loc), _np(aw))
        out.copy_(torch.from_numpy(r).to(out.dtype))

    def backward(value, shapes, lsi, loc, aw, grad_out, grad_value, grad_loc, grad_aw):
        gv, gl, ga = O.backward(_np(value), shapes.numpy(), lsi.numpy(), _np(loc), _np(aw), _np(grad_out))
        grad_value.copy_(torch.from_numpy(gv).to(grad_value.dtype))      # overwrite contract (ABI v4)
        grad_loc.copy_(torch.from_numpy(gl).to(grad_loc.dtype))
        grad_aw.copy_(torch.from_numpy(ga).to(grad_aw.dtype))

    def _clips(value, clips):
        G = value.shape[0]
        T = G // clips
        return [(c * T, (c + 1) * T) for c in range(clips)]

    def temporal_forward(value, shapes, lsi, ftab, loc_c, aw_c, loc_t, aw_t, clips, out):
        for a, b in _clips(value, clips):
            r = temporal_reference(_np(value[a:b]), shapes.numpy(), lsi.numpy(), ftab.numpy(), _np(loc_c[a:b]),
                                   _np(aw_c[a:b]), _np(loc_t[a:b]), _np(aw_t[a:b]))
            out[a:b].copy_(torch.from_numpy(r).to(out.dtype))

    def temporal_backward(value, shapes, lsi, ftab, loc_c, aw_c, loc_t, aw_t, grad_out, clips,
                          grad_value, gloc_c, gaw_c, gloc_t, gaw_t):
        for a, b in _clips(value, clips):
            r = temporal_reference(_np(value[a:b]), shapes.numpy(), lsi.numpy(), ftab.numpy(), _np(loc_c[a:b]),
                                   _np(aw_c[a:b]), _np(loc_t[a:b]), _np(aw_t[a:b]), _np(grad_out[a:b]))
            for dst, src in zip((grad_value, gloc_c, gaw_c, gloc_t, gaw_t), r[1:]):
                dst[a:b].copy_(torch.from_numpy(src).to(dst.dtype))

    monkeypatch.setattr(F, "_check_inputs", _check_inputs)
    monkeypatch.setattr(_native, "forward", forward)
    monkeypatch.setattr(_native, "backward", backward)
    monkeypatch.setattr(_native, "temporal_forward", temporal_forward)
    monkeypatch.setattr(_native, "temporal_backward", temporal_backward)

    def mask_rows(rows, mask, row_elems):                 # contract of msda_mask_rows: masked rows -> 0, in place
        rows[:, :row_elems].masked_fill_(mask.reshape(-1, 1), 0.0)

    monkeypatch.setattr(_native, "mask_rows", mask_rows)
