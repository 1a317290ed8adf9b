"""Host side of the operator: same names and contracts as the reference's
``src/models/ops/functions/ms_deform_attn_func.py``, calling the HIP library through the C ABI.

* ``MSDeformAttnFunction``          -- drop-in for the reference autograd.Function (:21-38).
* ``MSDeformAttnTemporalFunction``  -- fused current+temporal call used by the temporal modules.
* ``ms_deform_attn_core_pytorch``   -- the reference's "for debug and test only" pure-PyTorch function
  (:102-122), kept as part of the import surface.  The operator never routes through it.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _native


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _check_inputs(named):
    # same preconditions as ms_deform_attn_cuda.cu:28-38 (contiguity, device) -- RuntimeError like
    # AT_ASSERTM; CPU tensors fail the way ms_deform_attn.h:38,60 does.
    for name, t in named:
        _require(isinstance(t, torch.Tensor), "%s must be a tensor" % name)
        if not t.is_cuda:
            raise RuntimeError("Not implemented on the CPU (%s is not a GPU tensor)" % name)
        if name == "value" and t.dim() == 4:
            _native.value_strides(t)        # dense or head-major (include/msda.h value_strides); raises otherwise
        else:
            _require(t.is_contiguous(), "%s tensor has to be contiguous" % name)
    dev = named[0][1].device
    for name, t in named:
        _require(t.device == dev, "%s must be on the same device as value" % name)


def _check_op_shapes(value, shapes, lsi, loc, aw):
    _require(value.dim() == 4 and loc.dim() == 6 and aw.dim() == 5, "bad ranks for value/loc/attn")
    N, S, M, D = value.shape
    N2, Lq, M2, L, P, two = loc.shape
    _require((N2, M2, two) == (N, M, 2), "sampling_locations does not match value")
    _require(tuple(aw.shape) == (N, Lq, M, L, P), "attention_weights does not match sampling_locations")
    _require(shapes.dtype == torch.int64 and lsi.dtype == torch.int64,
             "spatial_shapes / level_start_index must be int64")
    _require(tuple(shapes.shape) == (L, 2) and tuple(lsi.shape) == (L,), "bad spatial_shapes/level_start_index")
    _require(aw.dtype == loc.dtype, "sampling_locations / attention_weights must share one dtype")
    _native.type_code(value.dtype, loc.dtype)       # value's dtype, or float32 beside a 16-bit value (raises otherwise)


def _im2col_step(batch, im2col_step):
    # ms_deform_attn_cuda.cu:50-52
    step = min(batch, int(im2col_step))
    _require(step > 0 and batch % step == 0, "batch(%d) must divide im2col_step(%d)" % (batch, step))
    return step


class MSDeformAttnFunction(Function):
    """Same call contract as the reference (``ms_deform_attn_func.py:21-38``):
    ``apply(value[N,S,M,D], spatial_shapes[L,2] i64, level_start_index[L] i64,
    sampling_locations[N,Lq,M,L,P,2], attention_weights[N,Lq,M,L,P], im2col_step) -> [N,Lq,M*D]``;
    gradients for arguments 0, 3 and 4 only; ``once_differentiable``."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step, padding_mask=None):
        # padding_mask (not in the reference's signature; used by MSDeformAttn only): the bool [N, S] mask `value`
        # was produced under.  The backward then zeroes the masked rows of ITS grad_value (the gradient of ref
        # ms_deform_attn.py:102-103's masked_fill) before returning it, and project_value's backward skips that pass.
        ctx.padding_mask = None
        if padding_mask is not None:
            _require(padding_mask.dtype == torch.bool and padding_mask.device == value.device and
                     padding_mask.numel() == value.shape[0] * value.shape[1], "padding_mask must be a bool [N, S] tensor on value's device")
            ctx.padding_mask = padding_mask.reshape(-1).contiguous()
        _check_inputs([("value", value), ("spatial_shapes", value_spatial_shapes),
                       ("level_start_index", value_level_start_index),
                       ("sampling_loc", sampling_locations), ("attn_weight", attention_weights)])
        _check_op_shapes(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                         attention_weights)
        N, S, M, D = value.shape
        Lq = sampling_locations.shape[1]
        ctx.im2col_step = im2col_step
        output = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
        if N > 0 and Lq > 0:
            step = _im2col_step(N, im2col_step)
            # the reference launches once per chunk of `step` batch rows (cu:61-75); result-neutral
            for n in range(0, N, step):
                _native.forward(value[n:n + step], value_spatial_shapes, value_level_start_index,
                                sampling_locations[n:n + step], attention_weights[n:n + step],
                                output[n:n + step])
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, aw = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        N = value.shape[0]
        # all three are fully written by the library (ABI v4: no zeros_like as in cu:121; skipped points -> 0).  16-bit
        # storage: grad_value comes back in the storage type where the library can write it so (ABI v10), else in fp32
        live = N > 0 and loc.shape[1] > 0
        step = _im2col_step(N, ctx.im2col_step) if live else 0
        acc = _native.grad_value_dtype(value[:step], shapes, loc.shape[1], loc.shape[3], loc.shape[4], grad_out=grad_output) if live else value.dtype
        grad_value = (torch.empty if live else torch.zeros)(value.shape, dtype=acc, device=value.device)
        grad_loc = torch.empty_like(loc)
        grad_aw = torch.empty_like(aw)
        if live:
            for n in range(0, N, step):
                _native.backward(value[n:n + step], shapes, lsi, loc[n:n + step], aw[n:n + step],
                                 grad_output[n:n + step], grad_value[n:n + step],
                                 grad_loc[n:n + step], grad_aw[n:n + step])
        if acc != value.dtype:
            grad_value = grad_value.to(value.dtype)
        if ctx.padding_mask is not None and grad_value.numel():
            N, S, M, D = grad_value.shape
            _native.mask_rows(grad_value.view(N * S, M * D), ctx.padding_mask, M * D)
        # one gradient slot per forward argument INCLUDING the optional padding_mask: autograd accepts trailing None
        # gradients beyond the inputs apply() was given, so the 6-argument (reference) call works with the same tuple
        return grad_value, None, None, grad_loc, grad_aw, None, None


class MSDeformAttnTemporalFunction(Function):
    """Fused form of the per-frame loop of the temporal modules
    (``ms_deform_attn.py:325-364, 366-404, 435-460``): for each frame t of each clip

        out[t] = MSDeformAttn(value[t], loc_curr[t], aw_curr[t])
               + MSDeformAttn(cat_w value[frame_table[t, w]], loc_temp[t], aw_temp[t])

    in ONE launch over ``value [clips*T, S, M, D]``, with no ``value[temporal_frames]`` copies.

    ``apply(value, spatial_shapes[L,2], level_start_index[L], frame_table[T,W] int32,
    loc_curr[G,Lq,M,L,Pc,2], aw_curr[G,Lq,M,L,Pc], loc_temp[G,Lq,M,W*L,Pt,2], aw_temp[G,Lq,M,W*L,Pt],
    clips) -> [G, Lq, M*D]`` with ``G = clips*T``."""

    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, frame_table, loc_curr, aw_curr,
                loc_temp, aw_temp, clips):
        _check_inputs([("value", value), ("spatial_shapes", spatial_shapes),
                       ("level_start_index", level_start_index), ("frame_table", frame_table),
                       ("loc_curr", loc_curr), ("aw_curr", aw_curr),
                       ("loc_temp", loc_temp), ("aw_temp", aw_temp)])
        _check_op_shapes(value, spatial_shapes, level_start_index, loc_curr, aw_curr)
        G, S, M, D = value.shape
        L = spatial_shapes.shape[0]
        Lq = loc_curr.shape[1]
        _require(clips > 0 and G % clips == 0, "value.shape[0] must be clips * frames")
        T = G // clips
        _require(frame_table.dtype == torch.int32 and frame_table.dim() == 2 and
                 frame_table.shape[0] == T, "frame_table must be int32 [frames, window]")
        W = frame_table.shape[1]
        _require(W > 0, "frame_table needs at least one temporal slot")
        _require(loc_temp.dim() == 6 and tuple(loc_temp.shape[:4]) == (G, Lq, M, W * L) and
                 loc_temp.shape[5] == 2, "loc_temp must be [G, Lq, M, window*L, Pt, 2]")
        _require(tuple(aw_temp.shape) == tuple(loc_temp.shape[:5]), "aw_temp does not match loc_temp")
        _require(loc_temp.dtype == loc_curr.dtype and aw_temp.dtype == loc_curr.dtype,
                 "current-frame and temporal sampling tensors must share one dtype")
        out = torch.empty((G, Lq, M * D), dtype=value.dtype, device=value.device)
        _native.temporal_forward(value, spatial_shapes, level_start_index, frame_table, loc_curr,
                                 aw_curr, loc_temp, aw_temp, clips, out)
        ctx.clips = clips
        ctx.save_for_backward(value, spatial_shapes, level_start_index, frame_table, loc_curr,
                              aw_curr, loc_temp, aw_temp)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, ftab, loc_c, aw_c, loc_t, aw_t = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        W = ftab.shape[1]
        acc = _native.grad_value_dtype(value, shapes, loc_c.shape[1], loc_c.shape[3], loc_c.shape[4], clips=ctx.clips,
                                       window=W, Pt=loc_t.shape[4], grad_out=grad_output)
        grad_value = torch.empty(value.shape, dtype=acc, device=value.device)      # overwritten (ABI v4)
        gloc_c, gaw_c = torch.empty_like(loc_c), torch.empty_like(aw_c)
        gloc_t, gaw_t = torch.empty_like(loc_t), torch.empty_like(aw_t)
        _native.temporal_backward(value, shapes, lsi, ftab, loc_c, aw_c, loc_t, aw_t, grad_output,
                                  ctx.clips, grad_value, gloc_c, gaw_c, gloc_t, gaw_t)
        if acc != value.dtype:
            grad_value = grad_value.to(value.dtype)
        return grad_value, None, None, None, gloc_c, gaw_c, gloc_t, gaw_t, None


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """The reference's debug/test helper (``ms_deform_attn_func.py:102-122``: "for debug and test
    only, need to use cuda version instead"), kept because ``functions/__init__.py`` exports it.

    Written here as an explicit bilinear gather in torch ops (no ``grid_sample``), so it also runs in
    bf16/fp16 and on any device.  NOT used by ``MSDeformAttnFunction`` or any module: those call the
    HIP kernels or raise.  Same signature, same result, same return layout ``[N, Lq, M*D]``."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in value_spatial_shapes.tolist()]
    out = value.new_zeros((N, Lq, M, D))
    n_idx = torch.arange(N, device=value.device).view(N, 1, 1, 1)
    m_idx = torch.arange(M, device=value.device).view(1, 1, M, 1)
    start = 0
    for lvl, (H, W) in enumerate(shapes):
        fmap = value[:, start:start + H * W].reshape(N, H, W, M, D)
        x = sampling_locations[:, :, :, lvl, :, 0] * W - 0.5      # [N, Lq, M, P]
        y = sampling_locations[:, :, :, lvl, :, 1] * H - 0.5
        x0, y0 = torch.floor(x), torch.floor(y)
        lx, ly = x - x0, y - y0
        a = attention_weights[:, :, :, lvl]
        for dy, wy in ((0, 1 - ly), (1, ly)):
            for dx, wx in ((0, 1 - lx), (1, lx)):
                xi, yi = (x0 + dx).long(), (y0 + dy).long()
                inside = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
                tap = fmap[n_idx, yi.clamp(0, H - 1), xi.clamp(0, W - 1), m_idx]   # [N,Lq,M,P,D]
                wgt = (wy * wx * a * inside.to(value.dtype)).unsqueeze(-1)
                out = out + (tap * wgt).sum(3)
        start += H * W
    return out.reshape(N, Lq, M * D).contiguous()


class _PaddedValueProj(Function):
    """``value_proj`` writing its output with a padded pixel stride.

    The reference views the Linear's dense ``[N*S, M*D]`` output as ``value[N, S, M, D]``; rows of one head
    are then exactly 1 KiB apart, and with one head per XCD (L2 locality) address bits 7..9 are constant on
    an XCD: its gathers use a fraction of the L2 channels (measured -17 % forward,
    -10 % gather pass when the bits vary).  A GEMM does not care about its output's leading dimension, so
    the product is written straight into a ``[N*S, (M + pad) * D]`` buffer (``ldc`` = padded row) and
    ``value`` is the ``[:, :, :M]`` view of it: same numbers, no extra pass, 1/M more memory.  The op takes
    the strided view through ``value_strides`` (include/msda.h); its ``grad_value`` comes back dense, so
    the backward GEMMs are the Linear's own."""

    @staticmethod
    def forward(ctx, x, weight, bias, n_heads, pad_heads, padding_mask, consumer_masks_grad):
        N, S, C = x.shape
        out_f = weight.shape[0]
        D = out_f // n_heads
        buf = torch.empty((N, S, n_heads + pad_heads, D), dtype=x.dtype, device=x.device)
        out2d = buf.view(N * S, (n_heads + pad_heads) * D)[:, :out_f]
        x2d = x.reshape(N * S, C)
        if bias is not None:
            torch.addmm(bias, x2d, weight.t(), out=out2d)
        else:
            torch.mm(x2d, weight.t(), out=out2d)
        value = buf[:, :, :n_heads]
        if padding_mask is not None:
            # ref :102-103 on the same elements; writes the masked rows only (msda_mask_rows), not a pass over value
            padding_mask = padding_mask.reshape(N * S).contiguous()
            _native.mask_rows(buf.view(N * S, (n_heads + pad_heads) * D), padding_mask, out_f)
        ctx.save_for_backward(x, weight, padding_mask)
        ctx.has_bias = bias is not None
        ctx.consumer_masks_grad = consumer_masks_grad
        return value

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_value):
        x, weight, padding_mask = ctx.saved_tensors
        N, S, C = x.shape
        if padding_mask is not None and not ctx.consumer_masks_grad:
            # the gradient of masked_fill: masked rows get none.  Out of place -- an incoming gradient is not ours to
            # modify; MSDeformAttn avoids this pass by letting the operator mask the grad_value it produces
            grad_value = grad_value.masked_fill(padding_mask.view(N, S, 1, 1), 0.0)
        g2d = grad_value.reshape(N * S, weight.shape[0])
        x2d = x.reshape(N * S, C)
        grad_x = (g2d @ weight).view(N, S, C) if ctx.needs_input_grad[0] else None
        grad_w = _split_k_wgrad(g2d, x2d) if ctx.needs_input_grad[1] else None
        grad_b = g2d.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return grad_x, grad_w, grad_b, None, None, None, None


_BMM_F32_OUT = {}       # device type -> does torch.bmm(16-bit, 16-bit, out_dtype=float32) run there (probed at first use)


def _bmm_f32(a, b):
    """Batched product of 16-bit operands with FLOAT32 results (the GEMM's own fp32 accumulators written out unrounded), or None
    where this torch build has no such kernel for the device (``aten::bmm.dtype``: GPU backends only)."""
    kind = a.device.type
    if _BMM_F32_OUT.get(kind, True):
        try:
            out = torch.bmm(a, b, out_dtype=torch.float32)
            _BMM_F32_OUT[kind] = True
            return out
        except (NotImplementedError, RuntimeError, TypeError):
            _BMM_F32_OUT[kind] = False
    return None


def _split_k_wgrad(g2d, x2d, rows_per_split=1024):
    """``g2d.t() @ x2d`` for a LONG reduction (rows = every pixel of a clip, 28 920 at 360x640 x 6 frames) onto a small
    ``[out, in]`` result: the BLAS library runs that as 32 workgroups of one 32x64 tile each (0.109 ms for 256 x 256 in fp32,
    0.104 in bf16 on MI355X -- the largest single kernel of a decoder-layer step, profiles/r04_logs/module_kernels.txt).
    Cut into slices of ~``rows_per_split`` rows it is one batched product plus a sum over the slices: 0.067 / 0.047 ms
    (scripts/wgrad_probe.py).  For 16-bit tensors the partial products are taken in FLOAT32 (``bmm(..., out_dtype=float32)``: the
    GEMM's accumulators, unrounded), summed in float32 and rounded ONCE, exactly as a single GEMM does.  (Round 5 summed 16-bit
    partials in fp32, which changed nothing: ``bmm`` had already rounded each of the ~28 partials -- relative RMS error 2.35e-3
    against 1.66e-3 for the single GEMM)  Where torch has no fp32-output product for the device, the single GEMM is
    used: slower, never less accurate."""
    R = g2d.shape[0]
    k = R // rows_per_split
    if k < 4 or not g2d.is_contiguous() or not x2d.is_contiguous():
        return g2d.t() @ x2d
    r = R // k
    main = r * k
    a, b = g2d[:main].view(k, r, -1).transpose(1, 2), x2d[:main].view(k, r, -1)
    if g2d.dtype in (torch.bfloat16, torch.float16):
        parts = _bmm_f32(a, b)
        if parts is None:
            return g2d.t() @ x2d
        w = parts.sum(0)
        if main < R:
            w = w + torch.mm(g2d[main:].t().float(), x2d[main:].float())       # (< k rows: a sliver)
        return w.to(g2d.dtype)
    w = torch.bmm(a, b).sum(0)
    if main < R:
        w.addmm_(g2d[main:].t(), x2d[main:])
    return w


def project_value(x, linear, n_heads, padding_mask=None, pad_heads=1, consumer_masks_grad=False):
    """``linear(x)`` viewed as ``value[N, S, M, D]`` (masked like ref ms_deform_attn.py:101-103), stored with
    ``pad_heads`` spare head slots per pixel row -- see :class:`_PaddedValueProj`.  ``pad_heads=0`` is the
    reference's dense layout.  The mask is applied by ``msda_mask_rows`` (writes the masked rows only).
    ``consumer_masks_grad``: the caller promises that the gradient flowing back into ``value`` already has zero
    rows where the mask is set (``MSDeformAttnFunction`` given the same ``padding_mask``), so the backward skips it."""
    if pad_heads <= 0 and padding_mask is None:
        return linear(x).view(x.shape[0], x.shape[1], n_heads, linear.out_features // n_heads)
    weight, bias = linear.weight, linear.bias
    if x.is_cuda and torch.is_autocast_enabled("cuda"):
        # under torch.autocast an nn.Linear computes and returns the autocast dtype; the padded product (an out= GEMM autocast
        # does not see) does the same, so that `value` -- and with it the operator's storage type -- is 16-bit as the user asked
        dt = torch.get_autocast_dtype("cuda")
        x, weight, bias = x.to(dt), weight.to(dt), (bias.to(dt) if bias is not None else None)
    return _PaddedValueProj.apply(x, weight, bias, n_heads, max(pad_heads, 0), padding_mask,
                                  bool(consumer_masks_grad) and padding_mask is not None)


def _check_prep_inputs(y, named_refs, shapes, sampling_dtype=None):
    """Preconditions of the fused pre-op pass.  msda_prep_* reads the reference points AS THE SAMPLING DTYPE (``y``'s, or
    float32 with MSDA_*_LOC32) and ``spatial_shapes`` as int64 device memory through raw pointers, so a mismatch would
    not fail, it would reinterpret memory: reference points are cast here when they differ (DeVIS's
    ``get_reference_points`` builds them in fp32 whatever the model's dtype: with float32 sampling they are used
    unrounded), everything must sit on ``y``'s GPU."""
    if not y.is_cuda:
        raise RuntimeError("Not implemented on the CPU (the fused pre-op pass needs GPU tensors)")
    _native.dtype_code(y.dtype)
    _require(isinstance(shapes, torch.Tensor) and shapes.dtype == torch.int64 and shapes.device == y.device and
             shapes.dim() == 2 and shapes.shape[1] == 2 and shapes.is_contiguous(),
             "spatial_shapes must be a contiguous int64 [L, 2] tensor on the device of the query")
    out = []
    for name, ref in named_refs:
        if ref is None:
            out.append(None)
            continue
        _require(isinstance(ref, torch.Tensor) and ref.is_floating_point(), "%s must be a floating-point tensor" % name)
        _require(ref.device == y.device, "%s must be on the same device as the query" % name)
        _require(ref.shape[-1] in (2, 4), "Last dim of reference_points must be 2 or 4, but get %d instead." % ref.shape[-1])
        out.append(ref.to(sampling_dtype or y.dtype).contiguous())
    return out


def _sampling_dtype(raw_dtype, loc32):
    """dtype of the sampling locations / attention weights the fused pre-op pass writes: the Linear outputs' dtype, or
    float32 beside 16-bit Linears when `loc32` (include/msda.h MSDA_*_LOC32: a bf16 coordinate resolves only 0.16 px on an
    80-pixel-wide level; the pass computes in fp32 anyway and hands the locations over unrounded)."""
    return torch.float32 if (loc32 and raw_dtype in (torch.bfloat16, torch.float16)) else raw_dtype


class MSDeformPrepFunction(Function):
    """Joint softmax + sampling-location arithmetic of the (temporal) modules as one fused pass each way
    (include/msda.h msda_prep_forward/backward).

    ``apply(off_c [R,M,L,Pc,2], off_t [R,M,W*L,Pt,2] | None, logit_c [R,M,L*Pc], logit_t [R,M,W*L*Pt] | None,
    ref_c [R,L,d], ref_t [R,W*L,d] | None, spatial_shapes [L,2])`` with R = frames*queries rows ->
    ``(loc_c, loc_t, aw_c, aw_t)`` in the operator's layouts (``loc_t``/``aw_t`` are None without a
    temporal part).  Same arithmetic as ref ms_deform_attn.py:112-121 (locations) and :252-258 (softmax)."""

    @staticmethod
    def forward(ctx, off_c, off_t, logit_c, logit_t, ref_c, ref_t, shapes, loc32=False):
        R, M, L, Pc, _ = off_c.shape
        W = 0 if off_t is None else off_t.shape[2] // L
        Pt = 1 if off_t is None else off_t.shape[3]
        ctx.ref_dtypes = (ref_c.dtype if isinstance(ref_c, torch.Tensor) else None,
                          ref_t.dtype if isinstance(ref_t, torch.Tensor) else None)
        sdt = _sampling_dtype(off_c.dtype, loc32)
        ref_c, ref_t = _check_prep_inputs(off_c, (("reference_points", ref_c), ("temporal reference_points", ref_t if W else None)), shapes, sdt)
        _require(logit_c.dtype == off_c.dtype and logit_c.device == off_c.device, "offsets / logits must share dtype and device")
        off_c, logit_c = off_c.contiguous(), logit_c.contiguous()
        _require(tuple(ref_c.shape) == (R, L, ref_c.shape[-1]), "reference_points must be [rows, L, 2|4]")
        if W:
            _require(off_t.dtype == off_c.dtype and logit_t.dtype == off_c.dtype, "offsets / logits must share one dtype")
            _require(tuple(ref_t.shape) == (R, W * L, ref_t.shape[-1]) and ref_t.shape[-1] == ref_c.shape[-1],
                     "temporal reference_points must be [rows, window*L, 2|4]")
            off_t, logit_t = off_t.contiguous(), logit_t.contiguous()
        loc_c, aw_c = torch.empty_like(off_c, dtype=sdt), torch.empty((R, M, L, Pc), dtype=sdt, device=off_c.device)
        loc_t = torch.empty_like(off_t, dtype=sdt) if W else None
        aw_t = torch.empty((R, M, W * L, Pt), dtype=sdt, device=off_c.device) if W else None
        _native.prep_forward(off_c, off_t, logit_c, logit_t, ref_c, ref_t, shapes, R, M, L, W, Pc, Pt,
                             loc_c, loc_t, aw_c, aw_t)
        ctx.save_for_backward(aw_c, aw_t, ref_c, ref_t, shapes, off_c, off_t)
        ctx.dims = (R, M, L, W, Pc, Pt)
        return loc_c, loc_t, aw_c, aw_t

    @staticmethod
    @once_differentiable
    def backward(ctx, gloc_c, gloc_t, gaw_c, gaw_t):
        aw_c, aw_t, ref_c, ref_t, shapes, off_c, off_t = ctx.saved_tensors
        R, M, L, W, Pc, Pt = ctx.dims
        zeros = lambda like: torch.zeros_like(like, dtype=aw_c.dtype)        # (the sampling-side dtype: see _sampling_dtype)
        gloc_c = zeros(off_c) if gloc_c is None else gloc_c.contiguous()
        gaw_c = zeros(aw_c) if gaw_c is None else gaw_c.contiguous()
        if W:
            gloc_t = zeros(off_t) if gloc_t is None else gloc_t.contiguous()
            gaw_t = zeros(aw_t) if gaw_t is None else gaw_t.contiguous()
        goff_c, glogit_c = torch.empty_like(off_c), torch.empty((R, M, L * Pc), dtype=off_c.dtype, device=off_c.device)
        goff_t = torch.empty_like(off_t) if W else None
        glogit_t = torch.empty((R, M, W * L * Pt), dtype=off_c.dtype, device=off_c.device) if W else None
        _native.prep_backward(gloc_c, gloc_t, gaw_c, gaw_t, aw_c, aw_t, ref_c, ref_t, shapes, R, M, L, W, Pc, Pt,
                              goff_c, goff_t, glogit_c, glogit_t)

        def ref_grad(gloc, off, ref, P):
            # d loc / d ref: 1 on (x, y); boxes: d loc / d (w, h) = offsets / P * 0.5   (ref :112-121)
            g = gloc.sum((1, 3))                                       # over heads and points -> [R, levels, 2]
            if ref.shape[-1] == 2:
                return g
            return torch.cat((g, (gloc * (off / P * 0.5)).sum((1, 3))), -1)

        gref_c = ref_grad(gloc_c, off_c, ref_c, Pc).to(ctx.ref_dtypes[0]) if ctx.needs_input_grad[4] else None
        gref_t = ref_grad(gloc_t, off_t, ref_t, Pt).to(ctx.ref_dtypes[1]) if (W and ctx.needs_input_grad[5]) else None
        return goff_c, goff_t, glogit_c, glogit_t, gref_c, gref_t, None, None


class MSDeformPrepFusedFunction(Function):
    """:class:`MSDeformPrepFunction` reading the offsets and logits as column slices of ONE matrix
    ``y [R, M*L*Pc*2 + M*W*L*Pt*2 + M*L*Pc + M*W*L*Pt]`` -- the output of the module's four query-side Linears
    run as a single GEMM (columns in that order) -- and writing their gradients as one matrix, so that the
    Linears' backward is one dgrad and one wgrad GEMM as well.  ``apply(y, ref_c, ref_t | None, shapes, M, L, W,
    Pc, Pt) -> (loc_c [R,M,L,Pc,2], loc_t, aw_c [R,M,L,Pc], aw_t)``."""

    @staticmethod
    def _cols(M, L, W, Pc, Pt):
        n = [M * L * Pc * 2, M * W * L * Pt * 2, M * L * Pc, M * W * L * Pt]
        o = [0, n[0], n[0] + n[1], n[0] + n[1] + n[2]]
        return [(a, a + b) for a, b in zip(o, n)]

    @staticmethod
    def forward(ctx, y, ref_c, ref_t, shapes, M, L, W, Pc, Pt, loc32=False):
        R = y.shape[0]
        if y.stride(1) != 1:
            y = y.contiguous()
        cols = MSDeformPrepFusedFunction._cols(M, L, W, Pc, Pt)
        assert y.shape[1] == cols[3][1]
        v = [y[:, a:b] for a, b in cols]
        ctx.ref_dtypes = (ref_c.dtype if isinstance(ref_c, torch.Tensor) else None,
                          ref_t.dtype if isinstance(ref_t, torch.Tensor) else None)
        sdt = _sampling_dtype(y.dtype, loc32)
        ref_c, ref_t = _check_prep_inputs(y, (("reference_points", ref_c), ("temporal reference_points", ref_t if W else None)), shapes, sdt)
        _require(tuple(ref_c.shape) == (R, L, ref_c.shape[-1]), "reference_points must be [rows, L, 2|4]")
        if W:
            _require(tuple(ref_t.shape) == (R, W * L, ref_t.shape[-1]) and ref_t.shape[-1] == ref_c.shape[-1],
                     "temporal reference_points must be [rows, window*L, 2|4]")
        loc_c = torch.empty((R, M, L, Pc, 2), dtype=sdt, device=y.device)
        aw_c = torch.empty((R, M, L, Pc), dtype=sdt, device=y.device)
        loc_t = torch.empty((R, M, W * L, Pt, 2), dtype=sdt, device=y.device) if W else None
        aw_t = torch.empty((R, M, W * L, Pt), dtype=sdt, device=y.device) if W else None
        _native.prep_forward(v[0], v[1] if W else None, v[2], v[3] if W else None, ref_c, ref_t, shapes, R, M, L, W,
                             Pc, Pt if W else 1, loc_c, loc_t, aw_c, aw_t, ld=y.stride(0))
        ctx.save_for_backward(aw_c, aw_t, ref_c, ref_t, shapes, y)
        ctx.dims = (R, M, L, W, Pc, Pt)
        return loc_c, loc_t, aw_c, aw_t

    @staticmethod
    @once_differentiable
    def backward(ctx, gloc_c, gloc_t, gaw_c, gaw_t):
        aw_c, aw_t, ref_c, ref_t, shapes, y = ctx.saved_tensors
        R, M, L, W, Pc, Pt = ctx.dims
        cols = MSDeformPrepFusedFunction._cols(M, L, W, Pc, Pt)
        gloc_c = torch.zeros_like(aw_c).unsqueeze(-1).repeat(1, 1, 1, 1, 2) if gloc_c is None else gloc_c.contiguous()
        gaw_c = torch.zeros_like(aw_c) if gaw_c is None else gaw_c.contiguous()
        if W:
            gloc_t = torch.zeros_like(aw_t).unsqueeze(-1).repeat(1, 1, 1, 1, 2) if gloc_t is None else gloc_t.contiguous()
            gaw_t = torch.zeros_like(aw_t) if gaw_t is None else gaw_t.contiguous()
        gy = torch.empty((R, y.shape[1]), dtype=y.dtype, device=y.device)
        g = [gy[:, a:b] for a, b in cols]
        _native.prep_backward(gloc_c, gloc_t, gaw_c, gaw_t, aw_c, aw_t, ref_c, ref_t, shapes, R, M, L, W, Pc,
                              Pt if W else 1, g[0], g[1] if W else None, g[2], g[3] if W else None, ld=gy.stride(0))

        def ref_grad(gloc, off2d, ref, P):
            gsum = gloc.sum((1, 3))
            if ref.shape[-1] == 2:
                return gsum
            off = off2d.reshape(gloc.shape)
            return torch.cat((gsum, (gloc * (off / P * 0.5)).sum((1, 3))), -1)

        gref_c = ref_grad(gloc_c, y[:, cols[0][0]:cols[0][1]], ref_c, Pc).to(ctx.ref_dtypes[0]) if ctx.needs_input_grad[1] else None
        gref_t = ref_grad(gloc_t, y[:, cols[1][0]:cols[1][1]], ref_t, Pt).to(ctx.ref_dtypes[1]) if (W and ctx.needs_input_grad[2]) else None
        return gy, gref_c, gref_t, None, None, None, None, None, None, None
