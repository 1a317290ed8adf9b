"""TEST INFRASTRUCTURE ONLY -- Python face of the CPU oracle.

Two independent restatements of the reference arithmetic, both pinned by ``tests/test_oracle.py``
against golden vectors produced from the reference's own Python oracle
(``tests/golden/make_golden.py`` imports ``/root/reference`` in the build container only):

* ``forward`` / ``backward``: the plain-C loop restatement of the reference CUDA kernels
  (``oracle/msda_oracle.c``; follows ``src/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-299``
  and the host chunk loop ``ms_deform_attn_cuda.cu:20-153``), called through ctypes on numpy arrays.
* ``grid_sample_forward``: a torch restatement of the reference's pure-PyTorch path
  ``ms_deform_attn_core_pytorch`` (``src/models/ops/functions/ms_deform_attn_func.py:102-122``),
  differentiable with autograd.  This is what ``bench.py`` times as the CPU baseline.

Never imported by ``devis_amd``; the product path fails loudly without its HIP library instead.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmsda_oracle.so")
_lib = None


def build(force=False):
    """Compile oracle/msda_oracle.c with gcc (make).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("msda_oracle.c", "msda_oracle_body.inc", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "all"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _prep(value, shapes, lsi, loc, attn):
    dt = np.asarray(value).dtype
    if dt not in (np.float32, np.float64):
        raise TypeError("oracle runs in float32 or float64 only, got %s" % dt)
    value = np.ascontiguousarray(value, dtype=dt)
    loc = np.ascontiguousarray(loc, dtype=dt)
    attn = np.ascontiguousarray(attn, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    lsi = np.ascontiguousarray(lsi, dtype=np.int64)
    N, S, M, D = value.shape
    N2, Lq, M2, L, P, two = loc.shape
    assert (N2, M2, two) == (N, M, 2) and attn.shape == (N, Lq, M, L, P)
    assert shapes.shape == (L, 2) and lsi.shape == (L,)
    return dt, value, shapes, lsi, loc, attn, (N, S, M, D, L, Lq, P)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def forward(value, shapes, lsi, loc, attn):
    """out[N,Lq,M*D]; numpy in, numpy out; dtype follows ``value`` (f32 or f64)."""
    dt, value, shapes, lsi, loc, attn, dims = _prep(value, shapes, lsi, loc, attn)
    N, S, M, D, L, Lq, P = dims
    out = np.empty((N, Lq, M * D), dtype=dt)
    fn = getattr(_load(), "msda_oracle_forward_f32" if dt == np.float32 else "msda_oracle_forward_f64")
    fn.restype = None
    fn(_p(value), _p(shapes), _p(lsi), _p(loc), _p(attn),
       *[ctypes.c_int(x) for x in dims], _p(out))
    return out


def backward(value, shapes, lsi, loc, attn, grad_out):
    """(grad_value, grad_loc, grad_attn) with the shapes of value / loc / attn."""
    dt, value, shapes, lsi, loc, attn, dims = _prep(value, shapes, lsi, loc, attn)
    N, S, M, D, L, Lq, P = dims
    grad_out = np.ascontiguousarray(grad_out, dtype=dt).reshape(N, Lq, M * D)
    gv, gl, ga = np.empty_like(value), np.empty_like(loc), np.empty_like(attn)
    fn = getattr(_load(), "msda_oracle_backward_f32" if dt == np.float32 else "msda_oracle_backward_f64")
    fn.restype = None
    fn(_p(value), _p(shapes), _p(lsi), _p(loc), _p(attn), _p(grad_out),
       *[ctypes.c_int(x) for x in dims], _p(gv), _p(gl), _p(ga))
    return gv, gl, ga


def level_start_index(shapes):
    shapes = np.asarray(shapes, dtype=np.int64)
    hw = shapes[:, 0] * shapes[:, 1]
    return np.concatenate([[0], np.cumsum(hw)[:-1]]).astype(np.int64)


# ----------------------------------------------------------------------------------------------
# torch restatement of the reference's pure-PyTorch path (the CPU baseline bench.py times).
# ----------------------------------------------------------------------------------------------
def grid_sample_forward(value, shapes, loc, attn):
    """Same result as the reference's ms_deform_attn_core_pytorch (ms_deform_attn_func.py:102-122):
    per level, view the level's slab of ``value`` as an image batch [N*M, D, H, W], sample it with
    ``F.grid_sample(2*loc-1, bilinear, zeros, align_corners=False)``, weight and accumulate.
    torch tensors in/out; differentiable."""
    import torch.nn.functional as F

    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    hw = [(int(h), int(w)) for h, w in (shapes.tolist() if hasattr(shapes, "tolist") else shapes)]
    acc = value.new_zeros((N * M, D, Lq))
    start = 0
    for lvl, (H, W) in enumerate(hw):
        fmap = value[:, start:start + H * W].permute(0, 2, 3, 1).reshape(N * M, D, H, W)
        grid = (2.0 * loc[:, :, :, lvl] - 1.0).permute(0, 2, 1, 3, 4).reshape(N * M, Lq, P, 2)
        taps = F.grid_sample(fmap, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
        w = attn[:, :, :, lvl].permute(0, 2, 1, 3).reshape(N * M, 1, Lq, P)
        acc = acc + (taps * w).sum(-1)
        start += H * W
    return acc.view(N, M * D, Lq).transpose(1, 2).contiguous()
