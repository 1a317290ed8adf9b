"""ctypes binding of the C ABI in include/msda.h (libmsda_hip.so) -- the only door to the kernels.

There is NO fallback: if the library is missing and cannot be built, or a call fails, this raises.
torch is used here only as plumbing (device pointers, current stream).
"""
import ctypes
import os
import threading

import torch

from . import build as _build

MSDA_ABI_VERSION = 13
BWD_WORKSPACE_BYTES = 64
_DTYPE_CODE = {torch.float32: 0, torch.float64: 1, torch.bfloat16: 2, torch.float16: 3}

# every symbol include/msda.h declares (tests check the library exports each of them)
EXPORTED_SYMBOLS = (
    "msda_version", "msda_build_info", "msda_last_error", "msda_forward", "msda_backward",
    "msda_temporal_forward", "msda_temporal_backward", "msda_backward_workspace_bytes",
    "msda_prep_forward", "msda_prep_backward", "msda_reload_knobs", "msda_last_route", "msda_mask_rows", "msda_grad_value_dtype",
    "msda_route_key", "msda_pin_route", "msda_clear_routes", "msda_route_count",
)

_lib = None
_lock = threading.Lock()
_vp, _ci = ctypes.c_void_p, ctypes.c_int


def load():
    """Load (building first if needed) libmsda_hip.so; raises RuntimeError when impossible."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        try:
            path = _build.ensure()      # builds when missing / stale (file-locked); a stale library without hipcc warns
        except Exception as e:  # no hipcc on this box and no prebuilt library
            raise RuntimeError(
                "devis_amd: the HIP library %s is missing and could not be built (%s). "
                "There is no CPU fallback for MSDeformAttn." % (_build.lib_path(), e))
        lib = ctypes.CDLL(path)
        for name in EXPORTED_SYMBOLS:
            if not hasattr(lib, name):
                raise RuntimeError("devis_amd: %s does not export %s" % (path, name))
        lib.msda_version.restype = _ci
        lib.msda_last_error.restype = ctypes.c_char_p
        if lib.msda_version() != MSDA_ABI_VERSION:
            raise RuntimeError("devis_amd: ABI version mismatch (library %d, binding %d); rebuild with "
                               "python -m devis_amd.build --force" % (lib.msda_version(), MSDA_ABI_VERSION))
        lib.msda_build_info.restype = ctypes.c_char_p
        if b"timing_only=1" in lib.msda_build_info() and os.environ.get("MSDA_ENABLE_HOOKS") != "1":
            raise RuntimeError("devis_amd: %s is a TIMING-ONLY build (kernels that skip work, wrong results); it loads only "
                               "with MSDA_ENABLE_HOOKS=1" % path)
        lib.msda_forward.restype = _ci
        lib.msda_forward.argtypes = [_ci] + [_vp] * 5 + [_ci] * 7 + [_vp, _vp, _vp, _vp]
        lib.msda_backward.restype = _ci
        lib.msda_backward.argtypes = [_ci] + [_vp] * 6 + [_ci] * 7 + [_vp, _ci] + [_vp] * 3 + [ctypes.c_longlong, _vp, _vp, _vp]
        lib.msda_grad_value_dtype.restype = _ci
        lib.msda_grad_value_dtype.argtypes = [_ci] * 11 + [_vp]
        lib.msda_backward_workspace_bytes.restype = ctypes.c_longlong
        lib.msda_backward_workspace_bytes.argtypes = [_ci] * 4
        lib.msda_temporal_forward.restype = _ci
        lib.msda_temporal_forward.argtypes = [_ci] + [_vp] * 8 + [_ci] * 10 + [_vp, _vp, _vp, _vp]
        lib.msda_temporal_backward.restype = _ci
        lib.msda_temporal_backward.argtypes = [_ci] + [_vp] * 9 + [_ci] * 10 + [_vp, _ci] + [_vp] * 5 + [ctypes.c_longlong, _vp, _vp, _vp]
        lib.msda_prep_forward.restype = _ci
        lib.msda_prep_forward.argtypes = [_ci] + [_vp] * 7 + [ctypes.c_longlong] + [_ci] * 6 + [ctypes.c_longlong] + [_vp] * 5
        lib.msda_last_route.restype = ctypes.c_char_p
        lib.msda_reload_knobs.restype = None
        lib.msda_reload_knobs.argtypes = []
        lib.msda_prep_backward.restype = _ci
        lib.msda_prep_backward.argtypes = [_ci] + [_vp] * 9 + [ctypes.c_longlong] + [_ci] * 6 + [ctypes.c_longlong] + [_vp] * 5
        lib.msda_mask_rows.restype = _ci
        lib.msda_mask_rows.argtypes = [_ci, _vp, _vp] + [ctypes.c_longlong] * 3 + [_vp]
        lib.msda_route_key.restype = _ci
        lib.msda_route_key.argtypes = [_ci] * 12 + [_vp, ctypes.c_char_p, _ci]
        lib.msda_pin_route.restype = _ci
        lib.msda_pin_route.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        lib.msda_clear_routes.restype = None
        lib.msda_clear_routes.argtypes = []
        lib.msda_route_count.restype = _ci
        lib.msda_route_count.argtypes = []
        _lib = lib
        _load_shipped_routes()
    return _lib


def is_loaded():
    return _lib is not None


# ---- measured route table (include/msda.h, ABI v12) -----------------------------------------------------------------
ROUTES_FILE = os.path.join(os.path.dirname(os.path.realpath(__file__)), "routes.json")


def route_key(backward, dtype_code_, clips, frames, window, S, M, D, L, Lq, Pc, Pt, shapes):
    """Key of a call shape (msda_route_key).  ``shapes``: [[H, W], ...] host values (list / tensor)."""
    flat = [int(v) for hw in (shapes.tolist() if hasattr(shapes, "tolist") else shapes) for v in hw]
    arr = (ctypes.c_int64 * len(flat))(*flat)
    buf = ctypes.create_string_buffer(512)
    n = _lib_or_load().msda_route_key(int(bool(backward)), dtype_code_, clips, frames, window, S, M, D, L, Lq, Pc, Pt, arr, buf, 512)
    _check(n if n < 0 else 0, "msda_route_key")
    return buf.value.decode()


def pin_route(key, settings):
    """Pin ``settings`` (dict name -> int, or the "name=value ..." string; empty: unpin) for call shape ``key``."""
    if isinstance(settings, dict):
        settings = " ".join("%s=%d" % (k, int(v)) for k, v in sorted(settings.items()))
    _check(_lib_or_load().msda_pin_route(key.encode(), settings.encode()), "msda_pin_route")


def clear_routes():
    _lib_or_load().msda_clear_routes()


def route_count():
    return _lib_or_load().msda_route_count()


def _lib_or_load():
    return _lib if _lib is not None else load()


def _file_arch(doc):
    """gfx name a routes file was measured on: its "arch" field, else the gfxNNN inside its "device" string."""
    import re
    if doc.get("arch"):
        return str(doc["arch"])
    m = re.search(r"gfx[0-9a-f]+", str(doc.get("device", "")))
    return m.group(0) if m else None


def _running_arch():
    """gfx name of the current device, or None without a GPU (the CPU-only build container: nothing will run the routes)."""
    try:
        if torch.cuda.is_available():
            return torch.cuda.get_device_properties(torch.cuda.current_device()).gcnArchName.split(":")[0]
    except Exception:       # noqa: BLE001 -- a property torch does not have on this build: no check
        pass
    return None


def load_routes(path, check_device=True):
    """Pin every entry of a routes file ({"device": ..., "routes": {key: {name: value}}}); returns the number of entries pinned.
    A table measured on another architecture than the running GPU's is skipped with a warning (timings do not carry over)."""
    import json
    with open(path) as f:
        doc = json.load(f)
    table = doc.get("routes", {})
    if check_device:
        want, have = _file_arch(doc), _running_arch()
        if want and have and want != have:
            import warnings
            warnings.warn("devis_amd: %s was measured on %s, this GPU is %s: its %d pins are not loaded"
                          % (path, want, have, len(table)))
            return 0
    for key, settings in table.items():
        pin_route(key, settings)
    return len(table)


def _load_shipped_routes():
    """devis_amd/routes.json -- the table audited on MI355X (devis_amd.tune --audit) -- unless MSDA_ROUTES=0 (A/B runs against the
    fallback rules) or MSDA_ROUTES=/another/file.json."""
    which = os.environ.get("MSDA_ROUTES", "")
    if which == "0":
        return
    path = which or ROUTES_FILE
    if os.path.exists(path):
        try:
            load_routes(path)
        except Exception as e:  # a broken table must not take the operator down: the rules are the fallback
            import warnings
            warnings.warn("devis_amd: ignoring %s (%s)" % (path, e))


def reload_knobs():
    """Re-read the MSDA_* test / measurement knobs (honoured only with MSDA_ENABLE_HOOKS=1; include/msda.h)."""
    load().msda_reload_knobs()


def last_route():
    """Kernels launched by the last library call of this thread (include/msda.h msda_last_route)."""
    return load().msda_last_route().decode("utf-8", "replace")


def dtype_code(dtype):
    try:
        return _DTYPE_CODE[dtype]
    except KeyError:
        raise RuntimeError("devis_amd: unsupported dtype %s (float32/float64/bfloat16/float16)" % dtype)


_LOC32_CODE = {torch.bfloat16: 4, torch.float16: 5}      # include/msda.h MSDA_BF16_LOC32 / MSDA_F16_LOC32


def type_code(dtype, loc_dtype):
    """msda_dtype of a call whose value / out / grad_out are `dtype` and whose sampling_loc / attn_weight (and their
    gradients) are `loc_dtype`: the same type, or float32 beside a 16-bit `dtype` (ABI v11: unrounded sampling locations)."""
    if loc_dtype == dtype:
        return dtype_code(dtype)
    if loc_dtype == torch.float32 and dtype in _LOC32_CODE:
        return _LOC32_CODE[dtype]
    raise RuntimeError("devis_amd: sampling locations / attention weights must have value's dtype (or float32 beside a "
                       "16-bit value), got %s beside %s" % (loc_dtype, dtype))


def acc_dtype(dtype):
    """Arithmetic type of a storage dtype: what grad_value buffers hold unless :func:`grad_value_dtype` allows the storage type."""
    return torch.float64 if dtype == torch.float64 else torch.float32


_CODE_DTYPE = {v: k for k, v in _DTYPE_CODE.items()}


def grad_value_dtype(value, shapes, Lq, L, Pc, clips=None, window=0, Pt=1, grad_out=None):
    """torch dtype the grad_value buffer of a backward call on ``value [G, S, M, D]`` should have (include/msda.h
    msda_grad_value_dtype): the 16-bit storage type itself when the owner-computes scatter will write it -- no fp32
    buffer and no conversion pass -- else the arithmetic type.  The library answers from the shapes; what it can only see
    at the call -- 16-byte alignment of ``value`` / ``grad_out`` (a contiguous view at an odd storage offset) and value
    strides that are multiples of 8 elements -- is checked here, so that such a call takes the arithmetic type (and the
    generic kernels) instead of being refused."""
    G, S, M, D = value.shape
    clips = G if clips is None else clips
    if value.dtype not in (torch.bfloat16, torch.float16) or G == 0 or clips == 0:
        return acc_dtype(value.dtype)
    if value.data_ptr() % 16 or (grad_out is not None and grad_out.data_ptr() % 16):
        return acc_dtype(value.dtype)
    if not value.is_contiguous() and any(st % 8 for st in (value.stride(0), value.stride(1), value.stride(2))):
        return acc_dtype(value.dtype)
    code = load().msda_grad_value_dtype(dtype_code(value.dtype), clips, G // clips, window, S, M, D, L, Lq, Pc, Pt,
                                        shapes_hint(shapes))
    return _CODE_DTYPE[code]


def _check(rc, what):
    if rc != 0:
        msg = load().msda_last_error().decode("utf-8", "replace")
        raise RuntimeError("devis_amd: %s failed (status %d): %s" % (what, rc, msg))


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(device):
    """Context in which `device` is the current HIP device: a no-op when it already is (the usual case: one process
    per GPU) -- torch.cuda.device() costs two device switches per call otherwise."""
    return _NO_SWITCH if device.index == torch.cuda.current_device() else torch.cuda.device(device)


def _p(t):
    return None if t is None else t.data_ptr()


def value_strides(value, frames=1):
    """include/msda.h `value_strides` of a [G, S, M, D] tensor: None for the reference's dense layout, else
    a 3 x int64 host array {clip, head, pixel}.  Accepted: unit channel stride and (for the temporal op)
    frames of one clip evenly spaced by S pixels -- e.g. the head-major layout of :func:`head_major`.
    Anything else fails like the reference's contiguity assert (ms_deform_attn_cuda.cu:28)."""
    if value.is_contiguous():
        return None
    G, S, M, D = value.shape
    sb, sp, sh, sd = value.stride()
    if (D > 1 and sd != 1) or sp <= 0 or sh < 0 or sb < 0 or (frames > 1 and sb != S * sp):
        raise RuntimeError("value tensor has to be contiguous (or head-major, see devis_amd.head_major)")
    return (ctypes.c_int64 * 3)(frames * sb, sh, sp)


def head_major(value):
    """The same [G, S, M, D] tensor stored head-major ([M, G, S, D] memory): the layout the gather kernels
    read ~25 % faster (include/msda.h).  A per-head batched GEMM for value_proj writes it directly; this
    helper makes a copy."""
    return value.permute(2, 0, 1, 3).contiguous().permute(1, 2, 0, 3)


_shape_hints = {}       # (data_ptr, _version, device) -> (tensor kept alive, ctypes int64 array): host copies of spatial_shapes
_registered = {}        # the same, for tensors whose host values were GIVEN (register_host_values): never evicted, never read back
_hint_log = None        # while a list: every tensor shapes_hint() serves is appended (devis_amd.graphed learns which of a layer's
                        # integer arguments steer kernel selection)


def _hint_key(t):
    return (t.data_ptr(), t._version, t.device)


def register_host_values(tensor, values):
    """Tell the binding the host values of an integer device tensor it will be handed as ``spatial_shapes`` (or that
    :func:`known_host_values` will be asked about): the interned pyramids of ``devis_amd.patch_transformer`` are built from Python
    ints, so their host copy exists before the device one -- ``shapes_hint`` then never reads the device (the
    reference rebuilds ``spatial_shapes`` on every forward, deformable_transformer.py:74-87, and every new tensor cost one
    synchronising device-to-host copy here).  The tensor is kept alive; the caller must not change it in place."""
    flat = [int(v) for v in values]
    if len(flat) != tensor.numel():
        raise ValueError("register_host_values: %d values for a tensor of %d elements" % (len(flat), tensor.numel()))
    _registered[_hint_key(tensor)] = (tensor, (ctypes.c_int64 * len(flat))(*flat))


def known_host_values(t):
    """Host values of an integer tensor as a tuple WITHOUT touching the device, or None when they are not known (CPU tensors:
    read; device tensors: registered, or served by shapes_hint before)."""
    if not isinstance(t, torch.Tensor) or t.is_floating_point():
        return None
    if not t.is_cuda:
        return tuple(t.reshape(-1).tolist())
    for table in (_registered, _shape_hints):
        hit = table.get(_hint_key(t))
        if hit is not None and hit[0] is t:
            return tuple(hit[1])
    return None


def shapes_hint(shapes):
    """include/msda.h `spatial_shapes_host`: a host copy of the device tensor `spatial_shapes`, for kernel selection
    only.  Cached per tensor (the transformer hands the same tensor to every layer of every
    step), so the device-to-host copy -- the only synchronisation -- happens once per distinct tensor, and never for a tensor
    whose values were registered (``register_host_values``: the interned pyramids of ``devis_amd.patch_transformer``)."""
    if _hint_log is not None:
        _hint_log.append(shapes)
    key = _hint_key(shapes)
    hit = _registered.get(key) or _shape_hints.get(key)
    if hit is not None and hit[0] is shapes:
        return hit[1]
    if shapes.is_cuda and torch.cuda.is_current_stream_capturing():
        # a tensor first seen inside a HIP-graph capture (the reference's transformer rebuilds spatial_shapes on every forward,
        # deformable_transformer.py:87) cannot be read back without breaking the capture: no hint -- the library then selects
        # its kernels from the sizes alone (include/msda.h: the hint is optional); run the call once eagerly, or keep the
        # tensor alive across steps, to capture the routes the hint enables
        return None
    host = shapes.detach().to("cpu", torch.int64).reshape(-1).tolist()
    arr = (ctypes.c_int64 * len(host))(*host)
    if len(_shape_hints) > 64:
        _shape_hints.clear()
    _shape_hints[key] = (shapes, arr)
    return arr


class hint_log:
    """``with hint_log() as served:`` -- the tensors shapes_hint() was asked about inside the block (by identity)."""

    def __enter__(self):
        global _hint_log
        self._previous, _hint_log = _hint_log, []
        self.served = _hint_log
        return self.served

    def __exit__(self, *exc):
        global _hint_log
        _hint_log = self._previous
        return False


def forward(value, shapes, lsi, loc, aw, out):
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    with _on(value.device):
        rc = load().msda_forward(type_code(value.dtype, loc.dtype), _p(value), _p(shapes), _p(lsi), _p(loc), _p(aw),
                                 N, S, M, D, L, Lq, P, _p(out), value_strides(value), shapes_hint(shapes), _stream(value))
    _check(rc, "msda_forward")


def bwd_workspace(device, batch, num_query, num_heads, virtual_levels):
    """Device scratch for one backward call (include/msda.h), uninitialised: the library zeroes the ticket counters
    at its head and writes the culling records before it reads them (ABI v8 -- no memset launch here).  It is
    drawn from torch's caching allocator per call rather than kept in a table of our own: the allocator IS a
    per-(device, stream, size) cache, and unlike a private one it stays correct when two backward passes run on
    different streams."""
    n = load().msda_backward_workspace_bytes(batch, num_query, num_heads, virtual_levels)
    return torch.empty((n + 3) // 4, dtype=torch.int32, device=device)


def backward(value, shapes, lsi, loc, aw, grad_out, grad_value, grad_loc, grad_aw):
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    with _on(value.device):
        ws = bwd_workspace(value.device, N, Lq, M, L)
        rc = load().msda_backward(type_code(value.dtype, loc.dtype), _p(value), _p(shapes), _p(lsi), _p(loc), _p(aw),
                                  _p(grad_out), N, S, M, D, L, Lq, P,
                                  _p(grad_value), dtype_code(grad_value.dtype), _p(grad_loc), _p(grad_aw), _p(ws), ws.numel() * 4,
                                  value_strides(value), shapes_hint(shapes), _stream(value))
    _check(rc, "msda_backward")


def temporal_forward(value, shapes, lsi, ftab, loc_c, aw_c, loc_t, aw_t, clips, out):
    G, S, M, D = value.shape
    frames = G // clips
    _, Lq, _, L, Pc, _ = loc_c.shape
    window = ftab.shape[1] if ftab is not None else 0
    Pt = loc_t.shape[4] if window else 1
    with _on(value.device):
        rc = load().msda_temporal_forward(
            type_code(value.dtype, loc_c.dtype), _p(value), _p(shapes), _p(lsi), _p(ftab), _p(loc_c), _p(aw_c),
            _p(loc_t), _p(aw_t), clips, frames, window, S, M, D, L, Lq, Pc, Pt, _p(out),
            value_strides(value, frames), shapes_hint(shapes), _stream(value))
    _check(rc, "msda_temporal_forward")


def temporal_backward(value, shapes, lsi, ftab, loc_c, aw_c, loc_t, aw_t, grad_out, clips,
                      grad_value, gloc_c, gaw_c, gloc_t, gaw_t, workspace=None):
    G, S, M, D = value.shape
    frames = G // clips
    _, Lq, _, L, Pc, _ = loc_c.shape
    window = ftab.shape[1] if ftab is not None else 0
    Pt = loc_t.shape[4] if window else 1
    with _on(value.device):
        ws = workspace if workspace is not None else bwd_workspace(value.device, G, Lq, M, L * (1 + window))
        rc = load().msda_temporal_backward(
            type_code(value.dtype, loc_c.dtype), _p(value), _p(shapes), _p(lsi), _p(ftab), _p(loc_c), _p(aw_c),
            _p(loc_t), _p(aw_t), _p(grad_out), clips, frames, window, S, M, D, L, Lq, Pc, Pt,
            _p(grad_value), dtype_code(grad_value.dtype), _p(gloc_c), _p(gaw_c), _p(gloc_t), _p(gaw_t), _p(ws), ws.numel() * 4,
            value_strides(value, frames), shapes_hint(shapes), _stream(value))
    _check(rc, "msda_temporal_backward")


def prep_forward(off_c, off_t, logit_c, logit_t, ref_c, ref_t, shapes, rows, M, L, W, Pc, Pt, loc_c, loc_t, aw_c, aw_t, ld=0):
    """msda_prep_forward (include/msda.h): joint softmax + sampling locations in one pass."""
    with _on(off_c.device):
        rc = load().msda_prep_forward(type_code(off_c.dtype, loc_c.dtype), _p(off_c), _p(off_t), _p(logit_c), _p(logit_t), _p(ref_c),
                                      _p(ref_t), _p(shapes), rows, M, L, W, Pc, Pt, ref_c.shape[-1], ld,
                                      _p(loc_c), _p(loc_t), _p(aw_c), _p(aw_t), _stream(off_c))
    _check(rc, "msda_prep_forward")


def prep_backward(gloc_c, gloc_t, gaw_c, gaw_t, aw_c, aw_t, ref_c, ref_t, shapes, rows, M, L, W, Pc, Pt,
                  goff_c, goff_t, glogit_c, glogit_t, ld=0):
    with _on(gloc_c.device):
        rc = load().msda_prep_backward(type_code(goff_c.dtype, gloc_c.dtype), _p(gloc_c), _p(gloc_t), _p(gaw_c), _p(gaw_t), _p(aw_c),
                                       _p(aw_t), _p(ref_c), _p(ref_t), _p(shapes), rows, M, L, W, Pc, Pt, ref_c.shape[-1], ld,
                                       _p(goff_c), _p(goff_t), _p(glogit_c), _p(glogit_t), _stream(gloc_c))
    _check(rc, "msda_prep_backward")


def mask_rows(rows, mask, row_elems):
    """In place: zero ``rows[i, :row_elems]`` of the 2-D (possibly row-padded) view ``rows`` wherever ``mask[i]``
    (include/msda.h msda_mask_rows; ref ms_deform_attn.py:102-103)."""
    if not rows.is_cuda:
        raise RuntimeError("Not implemented on the CPU (msda_mask_rows needs GPU tensors)")
    if not (rows.dim() == 2 and rows.stride(1) == 1 and rows.shape[1] >= row_elems and
            (rows.shape[0] <= 1 or rows.stride(0) >= row_elems)):
        raise ValueError("mask_rows: rows must be a 2-D view with contiguous rows")
    if not (mask.dtype == torch.bool and mask.device == rows.device and mask.is_contiguous() and mask.numel() == rows.shape[0]):
        raise ValueError("mask_rows: padding mask must be a contiguous bool tensor with one entry per row, on the rows' device")
    with _on(rows.device):
        rc = load().msda_mask_rows(dtype_code(rows.dtype), _p(rows), _p(mask), rows.shape[0], row_elems,
                                   rows.stride(0) if rows.shape[0] > 1 else max(row_elems, rows.stride(0)), _stream(rows))
    _check(rc, "msda_mask_rows")
