"""``MultiScaleDeformableAttention`` -- the module DeVIS's unmodified ``src/models/ops/functions/ms_deform_attn_func.py:18``
imports (``import MultiScaleDeformableAttention as MSDA``), as a ctypes binding of ``libmsda_hip.so`` (``include/msda.h``).

The reference builds a pybind11 extension of that name from ``src/models/ops/setup.py:36-71`` (``make.sh``) exposing the two
functions of ``src/vision.cpp:13-16``; this file is its MI355X counterpart: the same two functions with the same argument
lists, return values and error behaviour (``ms_deform_attn_cuda.cu:28-52``, ``ms_deform_attn.h:38,60``), over the C ABI.
Put it on the import path -- ``pip install .`` at the repository root installs it beside the ``devis_amd`` package
(``pyproject.toml`` / ``setup.py``), or copy it next to DeVIS's ``main.py`` -- and DeVIS's own Python runs unchanged
(path B).  ``devis_amd/_native.py`` + ``devis_amd/functions/ms_deform_attn_func.py`` are the fuller version of
the same binding (host shape hints, 16-bit ``grad_value``, fused temporal entry points); this one keeps to what the
reference's extension offers.

The library is looked for, in this order: ``$MSDA_LIB``; ``libmsda_hip.so`` inside an importable ``devis_amd`` package
(built on first use by ``devis_amd.build`` when hipcc is there); ``../devis_amd/libmsda_hip.so`` relative to this file (a
source checkout).  There is no CPU fallback.
"""
import ctypes
import os

import torch


def _library_path():
    env = os.environ.get("MSDA_LIB")
    if env:
        return env
    try:
        from devis_amd import build
        return build.ensure()
    except ImportError:
        pass
    here = os.path.dirname(os.path.realpath(__file__))
    path = os.path.join(os.path.dirname(here), "devis_amd", "libmsda_hip.so")
    if not os.path.exists(path):
        raise ImportError("MultiScaleDeformableAttention: libmsda_hip.so not found (set MSDA_LIB, or install devis_amd)")
    return path


_lib = ctypes.CDLL(_library_path())
_vp, _ci = ctypes.c_void_p, ctypes.c_int
# a library compiled with timing-only experiment kernels (include/msda.h msda_build_info) is never a production library
_lib.msda_build_info.restype = ctypes.c_char_p
if b"timing_only=1" in _lib.msda_build_info() and os.environ.get("MSDA_ENABLE_HOOKS") != "1":
    raise ImportError("MultiScaleDeformableAttention: %s is a TIMING-ONLY build of libmsda_hip.so (wrong results by construction)"
                      % _library_path())
_lib.msda_last_error.restype = ctypes.c_char_p
_lib.msda_forward.restype = _ci
_lib.msda_forward.argtypes = [_ci] + [_vp] * 5 + [_ci] * 7 + [_vp, _vp, _vp, _vp]
_lib.msda_backward.restype = _ci
_lib.msda_backward.argtypes = [_ci] + [_vp] * 6 + [_ci] * 7 + [_vp, _ci] + [_vp] * 3 + [ctypes.c_longlong, _vp, _vp, _vp]
_lib.msda_backward_workspace_bytes.restype = ctypes.c_longlong
_lib.msda_backward_workspace_bytes.argtypes = [_ci] * 4
_DT = {torch.float32: 0, torch.float64: 1, torch.bfloat16: 2, torch.float16: 3}      # include/msda.h msda_dtype


def _chk(rc):
    if rc:
        raise RuntimeError(_lib.msda_last_error().decode())


def _step(n, im2col_step):                                   # ms_deform_attn_cuda.cu:50-52
    s = min(n, im2col_step)
    if n % s:
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (n, s))
    return s


def _check_inputs(*tensors):                                 # ms_deform_attn_cuda.cu:28-38, ms_deform_attn.h:38,60
    names = ("value", "spatial_shapes", "level_start_index", "sampling_loc", "attn_weight", "grad_output")
    if not tensors[0].is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    for name, t in zip(names, tensors):
        if not t.is_contiguous():
            raise RuntimeError("%s tensor has to be contiguous" % name)
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % name)


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """``vision.cpp:14`` -> ``ms_deform_attn_cuda.cu:20-80``: [N, Lq, M*D]."""
    _check_inputs(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    out = torch.empty(N, Lq, M * D, dtype=value.dtype, device=value.device)
    if N == 0:
        return out
    with torch.cuda.device(value.device):
        st, step = torch.cuda.current_stream().cuda_stream, _step(N, im2col_step)
        for n in range(0, N, step):                          # the reference's chunk loop, cu:61-75
            _chk(_lib.msda_forward(_DT[value.dtype], value[n:].data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                   sampling_loc[n:].data_ptr(), attn_weight[n:].data_ptr(), step, S, M, D, L, Lq, P,
                                   out[n:].data_ptr(), None, None, st))     # None, None: dense value, no host shapes hint
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """``vision.cpp:15`` -> ``ms_deform_attn_cuda.cu:83-153``: [grad_value, grad_sampling_loc, grad_attn_weight]."""
    grad_output = grad_output.contiguous()
    _check_inputs(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    acc = torch.float64 if value.dtype == torch.float64 else torch.float32
    gv = torch.empty(value.shape, dtype=acc, device=value.device)          # fully overwritten (ABI v4; cu:121 zero-fills)
    gl, ga = torch.empty_like(sampling_loc), torch.empty_like(attn_weight)
    if N == 0:
        return [gv.to(value.dtype), gl, ga]
    with torch.cuda.device(value.device):
        st, step = torch.cuda.current_stream().cuda_stream, _step(N, im2col_step)
        nws = _lib.msda_backward_workspace_bytes(step, Lq, M, L)           # scratch: tickets + culling records
        for n in range(0, N, step):
            ws = torch.empty((nws + 3) // 4, dtype=torch.int32, device=value.device)   # uninitialised (ABI v8)
            _chk(_lib.msda_backward(_DT[value.dtype], value[n:].data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                    sampling_loc[n:].data_ptr(), attn_weight[n:].data_ptr(), grad_output[n:].data_ptr(),
                                    step, S, M, D, L, Lq, P, gv[n:].data_ptr(), _DT[acc], gl[n:].data_ptr(),
                                    ga[n:].data_ptr(), ws.data_ptr(), nws, None, None, st))
    return [gv.to(value.dtype), gl, ga]
