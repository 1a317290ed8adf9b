"""Multi-GPU forms of the temporal attention (one process per GPU, torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm).

Mode 1 -- clip-parallel replicas: independent clips per GPU, no collective inside the operator.  This is
what the reference does (DDP, 1 clip per GPU: main.py:131,142) and what bench.py --gpus N measures;
nothing in this file is needed for it.

Mode 2 -- one clip sharded over the ranks (BASELINE.json configs[3]).  Every output row (frame, query)
depends on its own sampling locations / weights and on the read-only value maps of ALL frames of the
clip, so:
  * the flattened pixel axis (T*S rows of `value`) is cut into `world` contiguous chunks: each rank runs
    value_proj on its chunk only, then ONE all-gather rebuilds value [T, S, M, D] on every rank
    (shard = T*S*C*e/world bytes: 3.7 MB fp32 for T=6 on the 360x640 pyramid at 8 ranks);
  * the queries of every frame are cut into `world` ranges: each rank runs the fused kernel on its own
    rows; outputs stay sharded (output_proj is row-local);
  * backward: each rank's kernel produces a partial grad_value for the whole clip; ONE reduce-scatter
    (sum) returns to every rank the gradient of its own chunk.
T = 6 frames on 8 GPUs does not divide, which is why the cut is on pixels/queries, not on frames.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): the all-gather of such small shards is
latency-bound (~tens of us), comparable to the kernel itself, so Mode 2 only pays when one clip's
per-layer work must be spread (large S, encoder attention) -- throughput scaling is Mode 1.
"""
import torch
import torch.distributed as dist
from torch.autograd import Function

from .functions import MSDeformAttnTemporalFunction


def shard_range(n, world, rank):
    """Balanced contiguous range [start, stop) of n items for `rank` of `world` (sizes differ by <= 1)."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def padded_chunk(n, world):
    """Rows per rank when n rows are padded up to a multiple of world (all-gather needs equal shards)."""
    return (n + world - 1) // world


class _AllGatherRows(Function):
    """forward: all-gather equal row-chunks [chunk, ...] -> [world*chunk, ...];
    backward: reduce-scatter (sum) of the gradient back to the owners."""

    @staticmethod
    def forward(ctx, chunk, group):
        ctx.group = group
        world = dist.get_world_size(group)
        chunk = chunk.contiguous()
        out = chunk.new_empty((world * chunk.shape[0],) + tuple(chunk.shape[1:]))
        dist.all_gather_into_tensor(out, chunk, group=group)
        return out

    @staticmethod
    def backward(ctx, grad):
        group = ctx.group
        world = dist.get_world_size(group)
        grad = grad.contiguous()
        rows = grad.shape[0] // world
        # the same call under RCCL (GPU) and gloo (the CPU tests: torch >= 2.x implements it there), so that the
        # world-size 2 / 3 / 8 tests execute the branch an 8-GPU node runs
        mine = grad.new_empty((rows,) + tuple(grad.shape[1:]))
        dist.reduce_scatter_tensor(mine, grad, op=dist.ReduceOp.SUM, group=group)
        return mine, None


class _StartGatherRows(Function):
    """:class:`_AllGatherRows` with BOTH collectives left in flight: forward issues the all-gather with ``async_op=True`` (RCCL
    runs it on its own stream) and parks the work handle in ``pending``; the gathered tensor must not be read before
    :class:`_FinishGatherRows` has waited on the handle.  backward: the reduce-scatter (sum) was ISSUED by
    :class:`_FinishGatherRows`'s backward -- which autograd runs right behind the kernel's -- and is only WAITED for here, where
    its result is consumed: whatever autograd runs in between (the backward of the next clip's kernel in a batch, of the layer's
    query-side GEMM and pre-op pass in the single-clip form) overlaps it."""

    @staticmethod
    def forward(ctx, chunk, group, pending, pending_bwd):
        ctx.group, ctx.pending_bwd = group, pending_bwd
        world = dist.get_world_size(group)
        chunk = chunk.contiguous()
        out = chunk.new_empty((world * chunk.shape[0],) + tuple(chunk.shape[1:]))
        pending.append(dist.all_gather_into_tensor(out, chunk, group=group, async_op=True))
        return out

    @staticmethod
    def backward(ctx, grad):
        if not ctx.pending_bwd:         # (the gradient did not pass through _FinishGatherRows: reduce here, synchronously)
            return _AllGatherRows.backward(ctx, grad) + (None, None)
        work, mine, keep = ctx.pending_bwd.pop()
        work.wait()
        del keep                        # the full-size gradient the collective was reading
        return mine, None, None, None


class _FinishGatherRows(Function):
    """Identity that makes the current stream wait for the all-gather :class:`_StartGatherRows` started; its backward starts
    the reduce-scatter of the gradient (``async_op=True``) and hands the work to :class:`_StartGatherRows`'s backward."""

    @staticmethod
    def forward(ctx, full, pending, pending_bwd, group):
        ctx.pending_bwd, ctx.group = pending_bwd, group
        while pending:
            pending.pop().wait()
        return full.view_as(full)

    @staticmethod
    def backward(ctx, grad):
        group = ctx.group
        world = dist.get_world_size(group)
        grad = grad.contiguous()
        mine = grad.new_empty((grad.shape[0] // world,) + tuple(grad.shape[1:]))
        work = dist.reduce_scatter_tensor(mine, grad, op=dist.ReduceOp.SUM, group=group, async_op=True)
        ctx.pending_bwd.append((work, mine, grad))
        return grad, None, None, None   # (a placeholder for the node in between: _StartGatherRows returns `mine`)


class PendingValue:
    """The all-gather of one layer's ``value`` in flight (see :func:`start_gather_value`)."""

    def __init__(self, full, pending, pending_bwd, group, n_frames, spatial_size):
        self._full, self._pending, self._pending_bwd, self._group = full, pending, pending_bwd, group
        self._shape = (n_frames, spatial_size)

    def wait(self):
        """The full ``[T, S, M, D]`` tensor, valid on the current stream from here on; differentiable."""
        full = _FinishGatherRows.apply(self._full, self._pending, self._pending_bwd, self._group)
        T, S = self._shape
        return full[: T * S].reshape((T, S) + tuple(full.shape[1:]))


def start_gather_value(value_chunk, n_frames, spatial_size, group=None):
    """Issue the all-gather of :func:`gather_value` WITHOUT waiting for it: whatever the caller enqueues next on its
    stream -- the layer's query-side GEMM and the fused pre-op pass (softmax + sampling locations), which do not
    depend on ``value`` -- runs while the shards travel over xGMI.  ``.wait()`` on the result returns the tensor."""
    pending, pending_bwd = [], []
    full = _StartGatherRows.apply(value_chunk, group, pending, pending_bwd)
    return PendingValue(full, pending, pending_bwd, group, n_frames, spatial_size)


_agreed = {}            # (ids and versions of the tensors, group) -> the tensors (kept alive: their identity is the key)


def check_ranks_agree(spatial_shapes, frame_table, group=None):
    """Every rank of a sharded clip must hold the SAME pyramid and frame table: each samples the gathered ``value`` with its own
    copy, and a mismatch (a rank fed another resize of the clip, another window) would not fail -- it would sample garbage.  The
    first time a (spatial_shapes, frame_table) pair is seen, its values are all-gathered (a few dozen bytes) and compared on the
    host: ONE collective and one synchronisation per new pair of tensors (none afterwards; interned tensors -- ``devis_amd.
    patch_transformer`` -- make that once per run).  Raises RuntimeError on EVERY rank when they differ."""
    key = (id(spatial_shapes), spatial_shapes._version, id(frame_table), frame_table._version, id(group))
    if key in _agreed:
        return
    mine = torch.cat([spatial_shapes.reshape(-1).to(torch.int64), frame_table.reshape(-1).to(torch.int64)])
    world = dist.get_world_size(group)
    sizes = mine.new_empty((world,))
    dist.all_gather_into_tensor(sizes, mine.new_tensor([mine.numel()]), group=group)
    if len(set(sizes.tolist())) != 1:
        raise RuntimeError("devis_amd.clip_parallel: the ranks of a sharded clip hold pyramids / frame tables of different sizes: %s"
                           % sizes.tolist())
    everyone = mine.new_empty((world * mine.numel(),))
    dist.all_gather_into_tensor(everyone, mine, group=group)
    rows = everyone.view(world, mine.numel()).tolist()
    if any(r != rows[0] for r in rows):
        bad = [i for i, r in enumerate(rows) if r != rows[0]]
        raise RuntimeError("devis_amd.clip_parallel: spatial_shapes / frame_table differ between the ranks of a sharded clip "
                           "(ranks %s against rank 0)" % bad)
    if len(_agreed) > 64:
        _agreed.clear()
    _agreed[key] = (spatial_shapes, frame_table)


def gather_value(value_chunk, n_frames, spatial_size, group=None):
    """value_chunk: this rank's [chunk, M, D] rows of the flattened [T*S, M, D] value tensor (chunk =
    padded_chunk(T*S, world); the last rank's tail rows beyond T*S are padding).  Returns the full
    [T, S, M, D] tensor on every rank; differentiable (reduce-scatter in backward)."""
    full = _AllGatherRows.apply(value_chunk, group)
    return full[: n_frames * spatial_size].reshape((n_frames, spatial_size) + tuple(value_chunk.shape[1:]))


def sharded_temporal_attention(value_chunk, n_frames, spatial_size, spatial_shapes, level_start_index,
                               frame_table, loc_curr, aw_curr, loc_temp, aw_temp, group=None, transport_dtype=None,
                               check_agreement=True):
    """Mode 2 for one clip.  value_chunk: this rank's rows of the flattened value (see gather_value);
    loc_*/aw_* hold this rank's query range of every frame ([T, Lq_local, M, ...]) -- tensors, or ONE callable that
    produces the four of them (``loc_curr``; the others None): it is called while the all-gather is in flight, so the
    layer's query-side GEMM and pre-op pass overlap the collective -- in both directions: the backward's reduce-scatter is
    issued behind the kernel's backward and waited for where its result is consumed.  Returns this rank's output rows
    [T, Lq_local, M*D].  One all-gather forward, one reduce-scatter backward.  ``check_agreement``: :func:`check_ranks_agree`
    (one small collective + synchronisation the first time a pair of tensors is seen).

    ``transport_dtype`` (torch.bfloat16 / torch.float16, with an fp32 model): ``value`` crosses xGMI -- and is sampled -- in that
    16-bit type, its gradient comes back through the reduce-scatter in it; sampling locations, attention weights and their
    gradients stay float32 (ABI v11 ``MSDA_*_LOC32``), the output is returned in float32.  Halves the bytes of both
    collectives (SURVEY f-3: value in the dtype the transport prefers) at the price of ``value`` rounded once to 16 bits
    (outputs within 5e-3 of the fp32 ones, tests/dist_worker.py)."""
    if check_agreement:
        check_ranks_agree(spatial_shapes, frame_table, group)
    out_dtype = None
    if transport_dtype is not None and value_chunk.dtype != transport_dtype:
        if value_chunk.dtype != torch.float32 or transport_dtype not in (torch.bfloat16, torch.float16):
            raise ValueError("transport_dtype: bfloat16 / float16 for a float32 value")
        out_dtype = value_chunk.dtype
        value_chunk = value_chunk.to(transport_dtype)
    if callable(loc_curr):
        pending = start_gather_value(value_chunk, n_frames, spatial_size, group)
        loc_curr, aw_curr, loc_temp, aw_temp = loc_curr()
        value = pending.wait()
    else:
        value = gather_value(value_chunk, n_frames, spatial_size, group)
    out = MSDeformAttnTemporalFunction.apply(value.contiguous(), spatial_shapes, level_start_index,
                                             frame_table, loc_curr.contiguous(), aw_curr.contiguous(),
                                             loc_temp.contiguous(), aw_temp.contiguous(), 1)
    return out if out_dtype is None else out.to(out_dtype)



def sharded_temporal_attention_batch(clips, n_frames, spatial_size, spatial_shapes, level_start_index, frame_table,
                                     group=None, transport_dtype=None, check_agreement=True):
    """Mode 2 for SEVERAL clips at once (a training step's batch, bench.py --mode sharded): every clip's all-gather is
    issued before the first kernel, so the collectives of clips 1.. travel over xGMI while the kernels of the clips
    before them run -- xGMI is point-to-point and a ring all-gather of one clip's ``value`` is bound by one link, so
    the next clip's shards are the cheapest thing to overlap it with.  ``clips``: a list of
    ``(value_chunk, loc_curr, aw_curr, loc_temp, aw_temp)`` as :func:`sharded_temporal_attention` takes them.
    Returns the list of this rank's output rows, one ``[T, Lq_local, M*D]`` per clip.  The BACKWARD collectives overlap the
    same way (round 6): autograd runs a clip's kernel backward, then ``_FinishGatherRows.backward`` issues that clip's
    reduce-scatter with ``async_op=True``, then the previous clip's kernel backward -- while the reduce-scatter travels -- and
    the results are waited for at the end, where the value_proj gradients consume them."""
    if check_agreement:
        check_ranks_agree(spatial_shapes, frame_table, group)
    pending, out_dtypes = [], []
    for value_chunk, *_ in clips:
        out_dtype = None
        if transport_dtype is not None and value_chunk.dtype != transport_dtype:
            if value_chunk.dtype != torch.float32 or transport_dtype not in (torch.bfloat16, torch.float16):
                raise ValueError("transport_dtype: bfloat16 / float16 for a float32 value")
            out_dtype = value_chunk.dtype
            value_chunk = value_chunk.to(transport_dtype)
        out_dtypes.append(out_dtype)
        pending.append(start_gather_value(value_chunk, n_frames, spatial_size, group))
    outs = []
    for (_, loc_curr, aw_curr, loc_temp, aw_temp), gathered, out_dtype in zip(clips, pending, out_dtypes):
        out = MSDeformAttnTemporalFunction.apply(gathered.wait().contiguous(), spatial_shapes, level_start_index, frame_table,
                                                 loc_curr.contiguous(), aw_curr.contiguous(), loc_temp.contiguous(),
                                                 aw_temp.contiguous(), 1)
        outs.append(out if out_dtype is None else out.to(out_dtype))
    return outs
