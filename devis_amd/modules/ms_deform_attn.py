"""Host-side mirror of the reference attention modules
(``/root/reference/src/models/ops/modules/ms_deform_attn.py``): same class names, constructor
signatures, sub-module names (= state-dict keys, so reference checkpoints load), attributes
(``im2col_step``), return contracts and parameter initialisation.

What differs is how the operator is driven: the temporal modules build the sampling locations of ALL
frames at once and make ONE fused call (``MSDeformAttnTemporalFunction``) instead of the reference's
Python loop of ``2*T`` operator calls plus ``T`` ``value[temporal_frames]`` gather copies
(ref ``:325-364``, ``:366-404``, ``:435-460``).  Setting ``module.fused = False`` replays the
reference's call pattern (``2*T`` calls of ``MSDeformAttnFunction``) -- same results, used by tests
and by the benchmark's "reference call pattern" line.
"""
import math
import threading
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

from .. import _native
from ..functions import (MSDeformAttnFunction, MSDeformAttnTemporalFunction, MSDeformPrepFunction,
                         MSDeformPrepFusedFunction, project_value)


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


def _check_heads(d_model, n_heads):
    # ref :40-48 / :157-165
    if d_model % n_heads != 0:
        raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
    if not _is_power_of_2(d_model // n_heads):
        warnings.warn("You'd better set d_model in MSDeformAttn to make the dimension of each attention "
                      "head a power of 2 which is more efficient in our CUDA implementation.")


def _ring_directions(n_heads):
    """Per-head unit directions on the L-inf ring: (cos, sin)(2*pi*m/M) / max(|cos|, |sin|) (ref :66-68)."""
    thetas = torch.arange(n_heads, dtype=torch.float32) * (2.0 * math.pi / n_heads)
    grid = torch.stack([thetas.cos(), thetas.sin()], -1)
    return grid / grid.abs().max(-1, keepdim=True)[0]


def _offset_bias(n_heads, mid_shape, n_points):
    """Bias of a sampling-offset Linear laid out [heads, *mid_shape, points, 2]: point i of every head
    starts (i+1) steps along the head's direction (ref :69-76, :177-199)."""
    steps = torch.arange(1, n_points + 1, dtype=torch.float32).view(n_points, 1)
    per_head = _ring_directions(n_heads).view(n_heads, 1, 2) * steps            # [M, P, 2]
    per_head = per_head.view((n_heads,) + (1,) * len(mid_shape) + (n_points, 2))
    return per_head.expand((n_heads,) + tuple(mid_shape) + (n_points, 2)).reshape(-1)


def _fused_linear_params(module, linears):
    """Weight / bias of several Linears on the same input as ONE [sum(out), in] matrix (a single GEMM forward, one dgrad
    and one wgrad GEMM backward).  While autograd is recording (training) the concatenation is part of the graph, so the
    gradients flow back to the individual parameters; without grad (inference: the call DeVIS's tracker times,
    tracker.py:320-323) the concatenated tensors are cached on the module and rebuilt when a parameter changed
    (``_version``) or moved (``data_ptr`` / dtype).  Writes through ``.data`` do not bump ``_version`` (the reference's own
    ``_reset_parameters`` initialises that way, EMA code updates that way): the cache is therefore also dropped by
    ``train()``, ``_apply()`` (``.to`` / ``.half`` / ``.cuda``), ``load_state_dict`` and ``_reset_parameters`` (``_FusedParamsCache``),
    and code that writes ``p.data`` of a live model calls ``module.invalidate_fused_params()``.  Under
    ``torch.inference_mode()`` nothing is cached (inference tensors could not be saved for a later backward)."""
    params = [q for lin in linears for q in (lin.weight, lin.bias)]
    if torch.is_grad_enabled() and any(q.requires_grad for q in params):
        return torch.cat([l.weight for l in linears]), torch.cat([l.bias for l in linears])
    if torch.is_inference_mode_enabled():
        return torch.cat([l.weight for l in linears]), torch.cat([l.bias for l in linears])
    if params[0].is_cuda and torch.cuda.is_current_stream_capturing():
        # inside a HIP-graph capture the concatenation is part of the graph: a cached copy would be baked in by address and go
        # stale (or be freed) at the next optimizer step
        return torch.cat([l.weight for l in linears]), torch.cat([l.bias for l in linears])
    key = tuple((q.data_ptr(), q._version, q.dtype) for q in params)
    cached = module.__dict__.get("_fused_params")
    if cached is None or cached[0] != key:
        with torch.no_grad():
            cached = (key, torch.cat([l.weight for l in linears]), torch.cat([l.bias for l in linears]))
        module.__dict__["_fused_params"] = cached
    return cached[1], cached[2]


class _FusedParamsCache:
    """Mixin: drops the cached concatenation of ``_fused_linear_params`` wherever parameters are rewritten without a
    ``_version`` bump or replaced."""

    def invalidate_fused_params(self):
        self.__dict__.pop("_fused_params", None)

    def train(self, mode=True):
        self.invalidate_fused_params()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_fused_params()
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_fused_params()
        return super()._load_from_state_dict(*args, **kwargs)


def _normalizer(spatial_shapes):
    """(W_l, H_l) per level: divides pixel offsets into normalised coordinates (ref :113-114)."""
    return torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)


def _locations(reference, offsets, normalizer, n_points):
    """Sampling locations from reference points broadcastable to [..., Lq, 1, L, 1, 2|4] and offsets
    [..., Lq, M, L, P, 2] (ref :112-121): 2-d refs add offsets in pixels of each level, 4-d refs (boxes)
    add offsets as a fraction of half the box size."""
    if reference.shape[-1] == 2:
        return reference + offsets / normalizer[None, None, None, :, None, :]
    if reference.shape[-1] == 4:
        return reference[..., :2] + offsets / n_points * reference[..., 2:] * 0.5
    raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
        reference.shape[-1]))


class MSDeformAttn(_FusedParamsCache, nn.Module):
    value_pad_heads = 1     # spare head slots per pixel row of `value` (0 = the reference's dense layout)
    fused_prep = True       # softmax + sampling-location arithmetic in one fused pass (False: torch ops)
    sampling_fp32 = True    # 16-bit modules: the fused pass hands sampling locations / weights to the operator in float32
                            # (a bf16 coordinate resolves 0.16 px on an 80-pixel level; False: the module's own dtype)

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        """Multi-Scale Deformable Attention Module (ref ``ms_deform_attn.py:30-132``).
        :param d_model      hidden dimension
        :param n_levels     number of feature levels
        :param n_heads      number of attention heads
        :param n_points     number of sampling points per attention head per feature level
        """
        super().__init__()
        _check_heads(d_model, n_heads)
        self.im2col_step = 64
        self.d_model = d_model
        self.n_levels = n_levels
        self.n_heads = n_heads
        self.n_points = n_points

        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        constant_(self.sampling_offsets.weight.data, 0.)
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(_offset_bias(self.n_heads, (self.n_levels,), self.n_points))
        constant_(self.attention_weights.weight.data, 0.)
        constant_(self.attention_weights.bias.data, 0.)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.)
        self.invalidate_fused_params()

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask):
        """
        :param query                    (N, Length_{query}, C)
        :param reference_points         (N, Length_{query}, n_levels, 2) in [0, 1], or (..., 4) boxes
        :param input_flatten            (N, sum_l H_l*W_l, C)
        :param input_spatial_shapes     (n_levels, 2) [(H_0, W_0), ...]
        :param input_level_start_index  (n_levels,)
        :param input_padding_mask       (N, sum_l H_l*W_l), True for padding elements
        :return (output (N, Length_{query}, C), None)
        """
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        # ref :96 (``assert (shapes[:, 0] * shapes[:, 1]).sum() == Len_in``: a device read per call there); here against the cached
        # host copy of the tensor -- one read per distinct ``spatial_shapes`` tensor, none inside a HIP-graph capture
        hint = _native.shapes_hint(input_spatial_shapes) if input_spatial_shapes.is_cuda else input_spatial_shapes.reshape(-1).tolist()
        if hint is not None:
            assert sum(int(hint[2 * l]) * int(hint[2 * l + 1]) for l in range(len(hint) // 2)) == Len_in
        M, L, P = self.n_heads, self.n_levels, self.n_points

        # value_proj writes value[N, S, M, D] with one spare head slot per pixel row (SURVEY f-3): same
        # numbers as ref :118-121, a layout the gather kernels read ~15 % faster; value_pad_heads = 0: dense
        value = project_value(input_flatten, self.value_proj, M, input_padding_mask, self.value_pad_heads,
                              consumer_masks_grad=True)         # the operator below zeroes grad_value's masked rows
        if reference_points.shape[-1] not in (2, 4):
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
                reference_points.shape[-1]))
        if self.fused_prep and query.is_cuda:
            # softmax + location arithmetic in one pass (SURVEY f-2); same numbers as the branch below
            R = N * Len_q
            y = F.linear(query.reshape(R, -1), *_fused_linear_params(self, (self.sampling_offsets, self.attention_weights)))
            locations, _, weights, _ = MSDeformPrepFusedFunction.apply(
                y, reference_points.reshape(R, L, reference_points.shape[-1]), None, input_spatial_shapes,
                M, L, 0, P, 1, self.sampling_fp32)
            locations, weights = locations.view(N, Len_q, M, L, P, 2), weights.view(N, Len_q, M, L, P)
        else:
            offsets = self.sampling_offsets(query).view(N, Len_q, M, L, P, 2)
            weights = F.softmax(self.attention_weights(query).view(N, Len_q, M, L * P), -1)
            weights = weights.view(N, Len_q, M, L, P)
            locations = _locations(reference_points[:, :, None, :, None, :], offsets,
                                   _normalizer(input_spatial_shapes), P)
        output = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index, locations.contiguous(),
                                            weights.contiguous(), self.im2col_step, input_padding_mask)
        output = self.output_proj(output)
        _flush_offset_checks(output)
        return output, None


class _FrameTables:
    """Frame tables of the temporal modules, cached per LIST of offset tensors, and the deferred range checks of device-side
    offsets.  One process-wide instance (``_FRAME_TABLES``): the encoder and the decoder stack of a model pass different
    lists (devis_transformer.py:103-121, 151-169), each keeps its own entry (a few entries, least recently used out); all
    state is guarded by a lock, since DeVIS's data-parallel wrapper and user code may run ``forward`` from several threads.

    The transformer hands the SAME offset tensors to every layer of one forward but builds NEW ones every forward, so a
    table is reused while the very same tensor objects come back (unchanged: ``_version``) and otherwise rebuilt WITHOUT a
    host synchronisation: a handful of tiny launches once per stack per forward.

    The reference indexes value[temporal_offsets[t] + t] with torch semantics: a negative index wraps once, anything else
    out of range trips the indexing kernel's device-side assert.  The kernels take absolute frame ids in [0, T), so the
    table is normalised here; the range check raises IndexError at once for CPU offsets and -- ``strict`` (module attribute
    ``STRICT_TEMPORAL_OFFSETS``, one synchronisation per NEW list of offsets) -- for device offsets; otherwise the verdict
    of a device-side check travels to pinned memory behind the work already queued and is raised by the first call that
    finds it arrived: the next ``_frame_table`` call of any temporal module, or the end of the module forward that used
    the table when the device happens to be done by then (``flush``).  A bad table is never served from the cache once its
    verdict is known, and it is wrapped into range either way: a bad offset reads a wrong frame, never memory outside the
    clip.  (torch._assert_async would abort the process on this ROCm build: no message, nothing to catch.)"""

    capacity = 8

    def __init__(self):
        self._lock = threading.Lock()
        self._entries = []          # [(offset tensors with versions, n_frames, device, table)], most recently used last
        self._pending = []          # [(pinned verdict, event, n_frames, offsets as given)]
        self._captured = {}         # id -> table served inside a HIP-graph capture: the graph holds its ADDRESS, so it is never
                                    # freed (a few hundred bytes per captured (layer stack, offsets list))

    @staticmethod
    def _describe(offsets):
        try:
            return " (temporal_offsets = %s)" % [o.detach().cpu().tolist() for o in offsets]
        except Exception:       # pragma: no cover -- never let the description mask the error
            return ""

    def raise_on_bad_offsets(self, wait=False):
        bad = None
        with self._lock:
            keep = []
            for host, done, n_frames, offsets in self._pending:
                if wait:
                    done.synchronize()
                if not done.query():
                    keep.append((host, done, n_frames, offsets))
                elif not bool(host) and bad is None:
                    bad = (n_frames, offsets)
            self._pending = keep
            if bad is not None:
                self._entries = [e for e in self._entries if not self._same(e[0], bad[1])]
        if bad is not None:
            raise IndexError("temporal_offsets point outside the clip's %d frames%s" % (bad[0], self._describe(bad[1])))

    flush = raise_on_bad_offsets

    @staticmethod
    def _same(held, offsets):
        return len(held) == len(offsets) and all(a is b and a._version == v for (a, v), b in zip(held, offsets))

    def get(self, temporal_offsets, n_frames, device):
        if not (torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing()):
            self.raise_on_bad_offsets()     # (polls events and may raise: not inside a HIP-graph capture, like _flush_offset_checks)
        with self._lock:
            for i, (held, nf, dev, table) in enumerate(self._entries):
                if nf == n_frames and dev == device and self._same(held, temporal_offsets):
                    self._entries.append(self._entries.pop(i))
                    if _capturing(table):
                        self._captured[id(table)] = table
                    return table
        table = torch.stack([o.to(device) for o in temporal_offsets]) \
            + torch.arange(n_frames, device=device)[:, None]
        in_range = ((table >= -n_frames) & (table < n_frames)).all()
        capturing = in_range.is_cuda and torch.cuda.is_current_stream_capturing()
        if not in_range.is_cuda or (STRICT_TEMPORAL_OFFSETS and not capturing):
            if not bool(in_range):
                raise IndexError("temporal_offsets point outside the clip's %d frames%s"
                                 % (n_frames, self._describe(temporal_offsets)))
        elif not capturing:
            host = torch.empty((), dtype=torch.bool, pin_memory=True)
            host.copy_(in_range, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            with self._lock:
                self._pending.append((host, done, n_frames, list(temporal_offsets)))
        table = torch.remainder(table, n_frames).to(torch.int32).contiguous()
        if _capturing(table):
            return table                # built INSIDE a capture: a temporary of that graph (its contents exist only after a replay)
        with self._lock:
            self._entries.append(([(o, o._version) for o in temporal_offsets], n_frames, device, table))
            del self._entries[:-self.capacity]
        return table


def _capturing(t):
    return t.is_cuda and torch.cuda.is_current_stream_capturing()


def _flush_offset_checks(like):
    """End of a temporal module's forward: raise for any range check whose verdict has reached the host by now (no wait)."""
    if like.is_cuda and torch.cuda.is_current_stream_capturing():
        return
    _FRAME_TABLES.flush()


STRICT_TEMPORAL_OFFSETS = False     # True: device-side temporal offsets are range-checked with one synchronisation per new list
_FRAME_TABLES = _FrameTables()


class TemporalMSDeformAttnBase(_FusedParamsCache, nn.Module):
    """Shared part of the temporal modules (ref ``:137-285``)."""

    fused = True   # False: replay the reference's 2*T-call pattern (same results)
    fused_prep = True       # joint softmax + sampling-location arithmetic in one fused pass (False: torch ops)
    sampling_fp32 = True    # 16-bit modules: float32 sampling locations / weights out of the fused pass (see MSDeformAttn)
    value_pad_heads = 1     # spare head slots per pixel row of `value` (0 = the reference's dense layout)

    def __init__(self, n_frames=36, d_model=256, n_levels=4, t_window=2, n_heads=8, n_curr_points=4,
                 n_temporal_points=2):
        """
        :param n_curr_points      sampling points per head per level in the query's own frame
        :param n_temporal_points  sampling points per head per level in each of the t_window other frames
        """
        super().__init__()
        _check_heads(d_model, n_heads)
        self.im2col_step = 64
        self.d_model = d_model
        self.n_frames = n_frames
        self.n_levels = n_levels
        self.t_window = t_window
        self.n_heads = n_heads
        self.n_curr_points = n_curr_points
        self.n_temporal_points = n_temporal_points

        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_curr_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_curr_points)
        self.temporal_sampling_offsets = nn.Linear(
            d_model, n_heads * n_levels * t_window * n_temporal_points * 2)
        self.temporal_attention_weights = nn.Linear(
            d_model, n_heads * n_levels * t_window * n_temporal_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        constant_(self.sampling_offsets.weight.data, 0.)
        constant_(self.temporal_sampling_offsets.weight.data, 0.)
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(
                _offset_bias(self.n_heads, (self.n_levels,), self.n_curr_points))
            self.temporal_sampling_offsets.bias.copy_(
                _offset_bias(self.n_heads, (self.t_window, self.n_levels), self.n_temporal_points))
        for lin in (self.attention_weights, self.temporal_attention_weights):
            constant_(lin.weight.data, 0.)
            constant_(lin.bias.data, 0.)
        for lin in (self.value_proj, self.output_proj):
            xavier_uniform_(lin.weight.data)
            constant_(lin.bias.data, 0.)
        self.invalidate_fused_params()

    def _compute_deformable_attention(self, query, input_flatten):
        """value [T,S,M,D]; current offsets [T,Lq,M,L,Pc,2]; temporal offsets [T,Lq,M,W*L,Pt,2]
        (slot-major, level-minor); attention weights from ONE softmax over the Pc*L + W*L*Pt logits of
        a (frame, query, head), split back into current [T,Lq,M,L,Pc] and temporal [T,Lq,M,W*L,Pt]
        (ref :225-266; note: no padding mask on value, ref :229-230)."""
        T, Len_q, _ = query.shape
        M, L, W = self.n_heads, self.n_levels, self.t_window
        Pc, Pt = self.n_curr_points, self.n_temporal_points
        value = project_value(input_flatten, self.value_proj, M, None, self._pad_heads(T))     # [T,S,M,D], padded rows
        temporal_offsets = self.temporal_sampling_offsets(query).view(T, Len_q, M, W * L, Pt, 2)
        logits = torch.cat([self.attention_weights(query).view(T, Len_q, M, L * Pc),
                            self.temporal_attention_weights(query).view(T, Len_q, M, W * L * Pt)], 3)
        weights = F.softmax(logits, -1)
        weights_curr = weights[..., :L * Pc].reshape(T, Len_q, M, L, Pc)
        weights_temporal = weights[..., L * Pc:].reshape(T, Len_q, M, W * L, Pt)
        curr_offsets = self.sampling_offsets(query).view(T, Len_q, M, L, Pc, 2)
        return value, curr_offsets, temporal_offsets, weights_curr, weights_temporal

    def _value_and_sampling(self, query, input_flatten, ref_curr, ref_temp, shapes):
        """value [T,S,M,D], loc_curr [T,Lq,M,L,Pc,2], loc_temp [T,Lq,M,W*L,Pt,2], w_curr [T,Lq,M,L,Pc],
        w_temp [T,Lq,M,W*L,Pt] from the query, reference points ``ref_curr [T,Lq,L,d]`` / ``ref_temp``
        (broadcastable to ``[T,Lq,W*L,d]``) and the current-frame ``shapes [L,2]`` (ref :225-266 followed by
        the location arithmetic of :327-352 / :436-452).  On the GPU the softmax and the location
        arithmetic are one fused pass each way (SURVEY f-2); ``fused_prep = False`` uses torch ops."""
        T, Len_q, _ = query.shape
        M, L, W = self.n_heads, self.n_levels, self.t_window
        Pc, Pt = self.n_curr_points, self.n_temporal_points
        d = ref_curr.shape[-1]
        if not (self.fused_prep and query.is_cuda):
            value, off_curr, off_temp, w_curr, w_temp = self._compute_deformable_attention(query, input_flatten)
            normalizer = _normalizer(shapes)
            loc_curr = _locations(ref_curr[:, :, None, :, None, :], off_curr, normalizer, Pc)
            loc_temp = _locations(ref_temp[:, :, None, :, None, :], off_temp, normalizer.repeat(W, 1), Pt)
            return value, loc_curr, loc_temp, w_curr, w_temp
        R = T * Len_q
        value = project_value(input_flatten, self.value_proj, M, None, self._pad_heads(T))
        # the four query-side Linears as ONE GEMM (same parameters, concatenated on the fly); the fused pass reads
        # its output as column slices and returns one gradient matrix, so their backward is one dgrad + one wgrad
        lins = (self.sampling_offsets, self.temporal_sampling_offsets, self.attention_weights,
                self.temporal_attention_weights)
        y = F.linear(query.reshape(R, -1), *_fused_linear_params(self, lins))
        loc_c, loc_t, w_c, w_t = MSDeformPrepFusedFunction.apply(
            y, ref_curr.reshape(R, L, d), ref_temp.expand(T, Len_q, W * L, d).reshape(R, W * L, d), shapes,
            M, L, W, Pc, Pt, self.sampling_fp32)
        return (value, loc_c.view(T, Len_q, M, L, Pc, 2), loc_t.view(T, Len_q, M, W * L, Pt, 2),
                w_c.view(T, Len_q, M, L, Pc), w_t.view(T, Len_q, M, W * L, Pt))

    def _pad_heads(self, n_frames):
        """Spare head slots of the padded `value` layout.  With 1 + frames*window >= 64 sources (e.g. a connect-all
        decoder over >= 9 frames) the scatter kernels do not apply and the one-kernel atomic backward only takes the
        reference's dense layout: padding would push the backward onto the slow generic kernel."""
        return 0 if 1 + n_frames * self.t_window > 63 else self.value_pad_heads

    @staticmethod
    def _raise_on_bad_offsets(wait=False):
        """Deferred range checks of device-side temporal offsets (see :class:`_FrameTables`): raises IndexError for the
        first one whose verdict has arrived (``wait``: synchronise on the outstanding ones first -- tests, debugging)."""
        _FRAME_TABLES.raise_on_bad_offsets(wait)

    @staticmethod
    def _frame_table(temporal_offsets, n_frames, device):
        """[T, W] absolute frame indices: frame_table[t] = temporal_offsets[t] + t (ref :339, :445); see
        :meth:`_FrameTables.get`."""
        return _FRAME_TABLES.get(temporal_offsets, n_frames, device)

    def _attend(self, value, shapes, level_start, temporal_offsets, loc_curr, w_curr, loc_temp, w_temp):
        """[T, Lq, C]: current-frame + temporal attention for every frame."""
        T = value.shape[0]
        if self.fused:
            table = self._frame_table(temporal_offsets, T, value.device)
            return MSDeformAttnTemporalFunction.apply(
                value, shapes[0], level_start[0], table, loc_curr.contiguous(),
                w_curr.contiguous(), loc_temp.contiguous(), w_temp.contiguous(), 1)
        # the reference's call pattern: per frame one current call and one call on the stacked frames
        frames = []
        for t in range(T):
            out_curr = MSDeformAttnFunction.apply(
                value[t][None].contiguous(), shapes[0], level_start[0], loc_curr[t][None].contiguous(),
                w_curr[t][None].contiguous(), self.im2col_step)
            stacked = value[temporal_offsets[t] + t].flatten(0, 1)[None]
            out_temp = MSDeformAttnFunction.apply(
                stacked.contiguous(), shapes[1], level_start[1], loc_temp[t][None].contiguous(),
                w_temp[t][None].contiguous(), self.im2col_step)
            frames.append(out_curr + out_temp)
        return torch.cat(frames, dim=0)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, temporal_offsets):
        raise NotImplementedError


class TemporalMSDeformAttnDecoder(TemporalMSDeformAttnBase):
    """Ref ``:288-414``.  Returns the 5-tuple ``(output [1, T*q, C], [T x current locations
    [1,q,M,L,Pc,2]], [T x temporal locations [1,q,M,W*L,Pt,2]], weights_curr [T,q,M,L,Pc],
    weights_temporal [T,q,M,W*L,Pt])`` that ``visualize_att_maps.py:158-169`` hooks."""

    def __init__(self, n_frames=36, d_model=256, n_levels=4, t_window=2, n_heads=8, n_curr_points=4,
                 n_temporal_points=2, dec_instance_aware_att=True):
        super().__init__(n_frames=n_frames, d_model=d_model, n_levels=n_levels, t_window=t_window,
                         n_heads=n_heads, n_curr_points=n_curr_points,
                         n_temporal_points=n_temporal_points)
        self.dec_instance_aware_att = dec_instance_aware_att

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, temporal_offsets):
        T = input_flatten.shape[0]
        per_frame = query.shape[1] // T
        query = query.reshape(T, per_frame, query.shape[-1])
        if reference_points.shape[0] != T:
            reference_points = reference_points.reshape((T, per_frame) + reference_points.shape[-2:])
        if reference_points.shape[-1] not in (2, 4):
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
                reference_points.shape[-1]))
        W = self.t_window
        if self.dec_instance_aware_att:
            # reference point of the SAME instance in each of the other frames (ref :342-344)
            table = self._frame_table(temporal_offsets, T, reference_points.device).long()
            ref_t = reference_points[table].permute(0, 2, 1, 3, 4).flatten(2, 3)      # [T, q, W*L, d]
        else:
            ref_t = reference_points.repeat(1, 1, W, 1)                               # ref :346-347
        value, loc_curr, loc_temp, w_curr, w_temp = self._value_and_sampling(
            query, input_flatten, reference_points, ref_t, input_spatial_shapes[0])

        output = self._attend(value, input_spatial_shapes, input_level_start_index, temporal_offsets,
                              loc_curr, w_curr, loc_temp, w_temp)
        output = self.output_proj(output.flatten(0, 1)[None])
        _flush_offset_checks(output)
        return (output, [loc_curr[t][None] for t in range(T)], [loc_temp[t][None] for t in range(T)],
                w_curr, w_temp)


class TemporalMSDeformAttnEncoder(TemporalMSDeformAttnBase):
    """Ref ``:417-464``: queries are the pixels themselves (Lq = S); temporal sampling in the other
    frames starts from the level-0 reference point (ref :447).  Returns ``(output [T, S, C], None)``."""

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, temporal_offsets):
        assert reference_points.shape[-1] == 2
        # temporal sampling in the other frames starts from the level-0 reference point (ref :447)
        value, loc_curr, loc_temp, w_curr, w_temp = self._value_and_sampling(
            query, input_flatten, reference_points, reference_points[:, :, :1], input_spatial_shapes[0])
        output = self._attend(value, input_spatial_shapes, input_level_start_index, temporal_offsets,
                              loc_curr, w_curr, loc_temp, w_temp)
        return self.output_proj(output), None
