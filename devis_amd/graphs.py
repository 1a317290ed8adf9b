"""HIP-graph wrapper for one attention layer (``devis_amd.graphed``).

One clip per GPU -- DeVIS's own setting (main.py:85; the call that matters at inference: tracker.py:320-323) -- is HOST-bound
when run eagerly: a ``TemporalMSDeformAttnDecoder`` layer step is ~0.5 ms of GPU time but 0.8-1.3 ms of wall clock (the BLAS
library alone spends 35-55 us of CPU per GEMM call).  The library only enqueues work (no allocation of its own, no
synchronisation), so a layer can be captured into a HIP graph as it is; this module does the bookkeeping:

    layer = devis_amd.graphed(model.transformer.decoder.layers[0].cross_attn, example_inputs)
    out = layer(query, reference_points, src, spatial_shapes, level_start_index, temporal_offsets)

* forward AND backward are captured (``torch.cuda.make_graphed_callables``): gradients flow to the floating-point inputs that
  required them at capture time and to the module's parameters, exactly as from the eager module;
* TRAINING calls (gradients wanted) must come from a NON-DEFAULT stream -- wrap the step in ``with devis_amd.graph_stream():``.
  On this PyTorch-ROCm build a backward graph replayed next to eager work on the legacy default stream stops writing its small
  outputs (bias and weight gradients) from the second replay on: they come back holding whatever the forward graph last left
  in that part of the pool, silently (``scripts/repro_graph_default_stream.py``: forty lines of pure torch, no code of this
  package; outputs and input gradients are right, any non-default stream is right).  A training call made on the default
  stream therefore runs the module EAGERLY (one warning; ``layer.eager_calls`` counts them); inference calls replay anywhere;
* one graph per SIGNATURE -- shapes / dtypes / ``requires_grad`` of the floating-point tensor arguments, and of the other
  arguments (the integer tensors ``spatial_shapes`` / ``level_start_index``, the list of ``temporal_offsets``, ``None`` masks):
  - an integer tensor the library reads a HOST COPY of (``spatial_shapes``: it chooses kernels, grids and LDS plans from it --
    found out during the warm-up call) counts by VALUE: the graph is only valid for the pyramid it was captured with, and any
    tensor holding that pyramid replays it.  The values come from the binding without touching the device when the tensor is an
    interned one of ``devis_amd.patch_transformer`` (which wraps the reference's ``prepare_data``, deformable_transformer.py:69-94)
    or was seen before; a brand-new tensor costs one device-to-host read, as in an eager call;
  - every other integer DEVICE tensor (``level_start_index``, the temporal copies of both, each frame's ``temporal_offsets`` --
    the reference's stacks rebuild them on every forward, devis_transformer.py:97-118,146-158) counts by shape and is a graph
    INPUT: its current contents are copied into the captured tensor before each replay (a few bytes, device to device), so the
    replay computes with the values of THIS call;
  - anything else (host tensors, Python values) counts by identity / value and is bound at capture.
  With ``patch_transformer`` in place the reference's own stack therefore replays ONE graph per layer step after step, with no
  host synchronisation (``tests/test_modules_gpu.py``: three training steps under ``torch.cuda.set_sync_debug_mode("error")``).
"""
import contextlib
import threading
import warnings

import torch
from torch import nn

_graph_streams = {}


@contextlib.contextmanager
def graph_stream(device=None):
    """Run the enclosed code on a non-default stream of ``device`` (one per device, kept), fenced against the default stream on
    entry and exit -- what graphed TRAINING steps need on this PyTorch-ROCm build (see the module docstring); nothing to do when
    the caller is on a non-default stream already."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    cur = torch.cuda.current_stream(device)
    if cur != torch.cuda.default_stream(device):        # already on a stream of the caller's own (or inside an outer graph_stream)
        yield cur
        return
    side = _graph_streams.get(device)
    if side is None:
        side = _graph_streams[device] = torch.cuda.Stream(device)
    side.wait_stream(cur)
    try:
        with torch.cuda.stream(side):
            yield side
    finally:
        cur.wait_stream(side)       # also when the body raised: what it queued on the side stream stays ordered before the caller's next work


def _is_flowing(x):
    return isinstance(x, torch.Tensor) and x.is_floating_point()


def _static_key(x):
    """Identity of a bound (non-flowing) argument: tensors by object and version, containers element-wise, the rest by value."""
    if isinstance(x, torch.Tensor):
        return ("t", id(x), x._version)
    if isinstance(x, (list, tuple)):
        return ("l", type(x).__name__) + tuple(_static_key(e) for e in x)
    return ("v", x)


def _int_leaves(x, path=()):
    """(path, tensor) of every integer DEVICE tensor inside a non-flowing argument (tuples / lists walked)."""
    if isinstance(x, torch.Tensor):
        if not x.is_floating_point() and x.is_cuda:
            yield path, x
    elif isinstance(x, (list, tuple)):
        for i, e in enumerate(x):
            yield from _int_leaves(e, path + (i,))


def _at(x, path):
    for i in path:
        x = x[i]
    return x


class _Bound(nn.Module):
    """The wrapped module with its non-flowing arguments fixed: what is captured (an nn.Module, so that its parameters are
    graph inputs and receive gradients)."""

    def __init__(self, inner, template, aux_grad):
        super().__init__()
        self.inner = inner
        self._template = template           # positional arguments; None at the places of the flowing tensors
        self._slots = [i for i, a in enumerate(template) if a is _FLOW]
        self._aux_grad = aux_grad
        self.none_slots = []

    def forward(self, *flowing):
        args = list(self._template)
        for i, t in zip(self._slots, flowing):
            args[i] = t
        out = self.inner(*args)
        if isinstance(out, tuple):
            # (a ``None`` among the outputs -- the temporal encoder's second one -- is not something a graph can return: left out
            # here, put back by GraphedLayer.__call__)
            self.none_slots = [i for i, o in enumerate(out) if o is None]
            out = tuple(o for o in out if o is not None)
        if self._aux_grad or not isinstance(out, tuple):
            return out
        # Only the FIRST output carries gradients through the graph: the temporal decoder's other four (sampling locations and
        # attention weights, returned for visualize_att_maps.py:158-169) are handed out detached -- as graph outputs that
        # require grad each of them would cost a zero-filled cotangent and a copy per backward replay.
        return (out[0],) + tuple(_detach(o) for o in out[1:])


def _detach(x):
    if isinstance(x, torch.Tensor):
        return x.detach()
    if isinstance(x, (list, tuple)):
        return type(x)(_detach(e) for e in x)
    return x


_FLOW = object()


class GraphedLayer:
    """Callable with the wrapped module's positional signature; see the module docstring.  ``graphs`` = number of captured
    signatures (tests, diagnostics)."""

    max_graphs = 16         # captured signatures kept (least recently captured out): a graph owns its static buffers, and a caller
                            # that hands over a NEW spatial_shapes tensor every step (see the module docstring) would pile them up

    def __init__(self, module, num_warmup_iters=3, aux_grad=False):
        self.module = module
        self.num_warmup_iters = num_warmup_iters
        self.aux_grad = aux_grad
        self._cache = {}
        self._keep = {}                     # signature -> the bound arguments (kept alive: the graph reads their memory)
        self._hinted = None                 # (argument index, path...) of the integer tensors the library reads a host copy of
        self._lock = threading.Lock()
        self.eager_calls = 0                # training calls made on the default stream (run eagerly: see the module docstring)

    @property
    def graphs(self):
        return len(self._cache)

    def _learn_hinted(self, args):
        """One eager call (no gradients): the library reads its host copy of ``spatial_shapes`` outside any capture, so that the
        graph records the routes that hint enables, and the binding tells which integer arguments it was asked about -- those
        steer kernel selection and count by value; the other integer device tensors become graph inputs."""
        from . import _native
        with torch.no_grad(), _native.hint_log() as served:
            self.module(*args)
        ids = {id(t) for t in served}
        self._hinted = {(i,) + path for i, a in enumerate(args) if not _is_flowing(a) for path, t in _int_leaves(a) if id(t) in ids}

    def _signature(self, args):
        from . import _native
        sig = []
        for i, a in enumerate(args):
            if _is_flowing(a):
                sig.append(("f", tuple(a.shape), a.dtype, a.device, a.requires_grad))
                continue
            leaves = dict(_int_leaves(a))
            if not leaves:
                sig.append(_static_key(a))
                continue

            def key(x, path):
                if isinstance(x, torch.Tensor) and path in leaves:
                    if (i,) + path in self._hinted:
                        vals = _native.known_host_values(x)
                        if vals is None:        # a tensor never seen before: one device read, as an eager call would make
                            vals = tuple(_native.shapes_hint(x))
                        return ("hv", tuple(x.shape), vals)
                    return ("in", tuple(x.shape), x.dtype, x.device)
                if isinstance(x, (list, tuple)):
                    return ("l", type(x).__name__) + tuple(key(e, path + (j,)) for j, e in enumerate(x))
                return _static_key(x)
            sig.append(key(a, ()))
        # (+ what else changes the captured kernels: train / eval, and the ambient autocast state -- under torch.autocast use
        # ``cache_enabled=False``, as torch.cuda.make_graphed_callables asks)
        autocast = (torch.get_autocast_dtype("cuda"),) if torch.is_autocast_enabled("cuda") else ()
        return tuple(sig) + (self.module.training,) + autocast

    def capture(self, *args):
        """Capture (or fetch) the graph for this signature without running it: call once per shape before timing."""
        if not any(_is_flowing(a) and a.is_cuda for a in args):
            raise RuntimeError("devis_amd.graphed: needs CUDA (HIP) tensor arguments -- there is no CPU path to capture")
        with self._lock:
            if self._hinted is None:
                self._learn_hinted(args)
            sig = self._signature(args)
            fn = self._cache.get(sig)
            if fn is not None:
                self._refresh_inputs(sig, args)
                return fn
            template = [_FLOW if _is_flowing(a) else a for a in args]
            flowing = tuple(a for a in args if _is_flowing(a))
            # one eager call first: the library reads its host copy of spatial_shapes (devis_amd._native.shapes_hint) outside the
            # capture, so that the graph records the routes that hint enables
            with torch.no_grad():
                self.module(*args)
            samples = tuple(t.detach().clone().requires_grad_(t.requires_grad) for t in flowing)
            bound = _Bound(self.module, template, self.aux_grad)
            # (the capture runs the backward too: grad mode on, even when the first call comes from inside torch.no_grad() --
            # inference, tracker.py:320-323 -- where the graphed callable then replays its forward graph only)
            with graph_stream(next(t.device for t in flowing if t.is_cuda)), torch.enable_grad():
                fn = torch.cuda.make_graphed_callables(bound, samples, num_warmup_iters=self.num_warmup_iters)
            self._cache[sig] = fn
            self._keep[sig] = list(args)        # (the flowing ones too: positions must line up; they are tiny references)
            while len(self._cache) > self.max_graphs:
                oldest = next(iter(self._cache))
                del self._cache[oldest], self._keep[oldest]
            return fn

    def _refresh_inputs(self, sig, args):
        """Before a replay: the integer device tensors that are graph inputs (module docstring) take this call's contents."""
        kept = self._keep[sig]
        for i, a in enumerate(args):
            if _is_flowing(a):
                continue
            for path, t in _int_leaves(a):
                if (i,) + path in self._hinted:
                    continue
                mine = _at(kept[i], path)
                if mine is not t:
                    mine.copy_(t, non_blocking=True)

    def _wants_gradients(self, args):
        return torch.is_grad_enabled() and (any(_is_flowing(a) and a.requires_grad for a in args) or
                                            any(p.requires_grad for p in self.module.parameters()))

    def __call__(self, *args):
        dev = next((a.device for a in args if _is_flowing(a) and a.is_cuda), None)
        if dev is not None and self._wants_gradients(args) and torch.cuda.current_stream(dev) == torch.cuda.default_stream(dev):
            if not self.eager_calls:
                warnings.warn("devis_amd.graphed: a call that needs gradients was made on the default stream and runs eagerly -- "
                              "on this PyTorch-ROCm build a backward graph replayed next to default-stream work returns stale "
                              "parameter gradients (scripts/repro_graph_default_stream.py); wrap the training step in "
                              "`with devis_amd.graph_stream():`", stacklevel=2)
            self.eager_calls += 1
            out = self.module(*args)
            if self.aux_grad or not isinstance(out, tuple):
                return out
            return (out[0],) + tuple(_detach(o) for o in out[1:])
        fn = self.capture(*args)
        with graph_stream(dev):             # (no-op inside a caller's own graph_stream; inference calls from the default stream replay here too)
            flowing = [a for a in args if _is_flowing(a)]
            for a in flowing:
                if a.is_cuda:
                    a.record_stream(torch.cuda.current_stream(dev))
            out = fn(*flowing)
        if not isinstance(out, tuple):
            return out
        if not self.aux_grad:
            # (the graphed autograd function marks every output as differentiable; the auxiliary ones carry no gradient inside
            # the graph -- _Bound.forward -- and say so here)
            out = (out[0],) + tuple(_detach(o) for o in out[1:])
        for i in getattr(fn, "none_slots", ()):
            out = out[:i] + (None,) + out[i:]
        return out


def graphed(module, example_inputs=None, num_warmup_iters=3, aux_grad=False):
    """Static-shape HIP-graph wrapper around one attention layer (``MSDeformAttn``, ``TemporalMSDeformAttnEncoder`` /
    ``Decoder``, or any module built on this package's operator): returns a :class:`GraphedLayer`.  ``example_inputs`` (the
    positional arguments of one call) are captured right away; other shapes are captured on first use.  Of a tuple of outputs
    only the first is differentiable through the graph unless ``aux_grad=True`` (the others are returned detached)."""
    layer = GraphedLayer(module, num_warmup_iters, aux_grad)
    if example_inputs is not None:
        layer.capture(*example_inputs)
    return layer
