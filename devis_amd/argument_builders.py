"""Cached form of the one call-site argument builder the reference exposes as a function:
``DeformableTransformerEncoder.get_reference_points`` (``src/models/deformable_transformer.py:185-198``), which the
DeVIS encoder stack calls once per forward (``devis_transformer.py:95``).  The reference rebuilds two ``linspace`` s and
a ``meshgrid`` per pyramid level per call -- each ``linspace`` reads its bounds from the device tensor
``spatial_shapes``, i.e. synchronises -- although the centre grids only depend on the pyramid.  Here the grids are
cached per pyramid (read from the host copy ``_native.shapes_hint`` already keeps per ``spatial_shapes`` tensor), and
only the per-call arithmetic with ``valid_ratios`` runs, in the reference's order: the result is bit-identical
(``tests/test_host_cpu.py::test_cached_reference_points_match_the_reference_call_site``, fixture made by the
reference).

``DeformableTransformer.prepare_data`` (``deformable_transformer.py:69-94``) is the other one: it has every ``(h, w)`` as Python
ints (``:74-75``), pushes them to the device as a NEW ``spatial_shapes`` tensor each forward (``:87``: a blocking copy) and derives
``level_start_index`` from it on the device.  ``patch_transformer`` replaces it by :func:`prepare_data`, which does the same
flattening and takes the two tensors from :func:`interned_pyramid` -- one device pair per pyramid and device, built from the
same Python ints and registered with the binding together with their host values (``_native.register_host_values``) -- so that
the operator's kernel selection never reads the device, and a ``devis_amd.graphed`` layer sees the same pyramid (by value) step
after step: one capture, no synchronisation.

Opt-in wiring, no DeVIS source change::

    import src.models.deformable_transformer as dt, src.models.devis_transformer as dvt
    devis_amd.patch_transformer(dt, dvt)

The other call-site tensors (temporal offsets, repeated shapes: ``devis_transformer.py:97-118,146-158``) are built inline in the
reference's ``forward`` s from the frame count alone.  With the second argument each frame's offsets tensor is interned as well
(:class:`_InterningTorch`); without it the attention modules consume new ones as they come
(``TemporalMSDeformAttnBase._frame_table``: one cached, synchronisation-free device table per list of offsets), and a
``devis_amd.graphed`` layer copies them into its captured arguments before each replay (a few bytes, device to device).
"""

import torch

_grid_cache = {}
_captured = {}
_MAX_ENTRIES = 32


def _pyramid(spatial_shapes):
    """[(H, W), ...] as Python ints.  A device tensor is read through the per-tensor host copy the operator keeps
    anyway (one device-to-host copy per distinct tensor)."""
    if isinstance(spatial_shapes, torch.Tensor):
        if spatial_shapes.is_cuda:
            from . import _native
            hint = _native.shapes_hint(spatial_shapes)
            if hint is None:
                raise RuntimeError("get_reference_points: spatial_shapes is a device tensor first seen inside a HIP-graph capture; "
                                   "call once outside the capture (or pass the sizes as Python ints)")
            flat = list(hint)
        else:
            flat = spatial_shapes.reshape(-1).tolist()
        return tuple((int(flat[2 * i]), int(flat[2 * i + 1])) for i in range(len(flat) // 2))
    return tuple((int(h), int(w)) for h, w in spatial_shapes)


def get_reference_points(spatial_shapes, valid_ratios, device):
    """Drop-in for the reference's static method (same arguments, same result, ``[N, S, L, 2]``): pixel centres of
    every level, normalised by the valid part of the (padded) frame."""
    shapes = _pyramid(spatial_shapes)
    key = (shapes, str(device))
    capturing = torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing()
    grids = _grid_cache.get(key)
    if grids is not None and capturing:
        _captured[id(grids)] = grids            # a HIP graph now holds these addresses: out of the cache's reach for good (keyed by
                                                # the grid set itself: after an eviction one key may have had several live sets)
    if grids is None:
        grids = []
        for H_, W_ in shapes:
            ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device),
                                          torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device),
                                          indexing='ij')
            grids.append((ref_y.reshape(-1)[None], ref_x.reshape(-1)[None]))
        if not capturing:                       # (grids built INSIDE a capture live in the graph's memory pool: never shared)
            if len(_grid_cache) >= _MAX_ENTRIES:
                _grid_cache.clear()
            _grid_cache[key] = grids
    per_level = []
    for lvl, ((H_, W_), (gy, gx)) in enumerate(zip(shapes, grids)):
        ref_y = gy / (valid_ratios[:, None, lvl, 1] * H_)
        ref_x = gx / (valid_ratios[:, None, lvl, 0] * W_)
        per_level.append(torch.stack((ref_x, ref_y), -1))
    points = torch.cat(per_level, 1)
    return points[:, :, None] * valid_ratios[:, None]


_pyramids = {}          # (shapes, device) -> (spatial_shapes, level_start_index): interned for good (a few hundred bytes per pyramid)


def interned_pyramid(shapes, device):
    """The device tensors ``(spatial_shapes [L, 2], level_start_index [L])`` (int64, as ``prepare_data`` builds them:
    deformable_transformer.py:87-91) of a pyramid given as Python ints -- ONE pair per (pyramid, device), built without reading
    the device and registered with the binding together with their host values."""
    shapes = tuple((int(h), int(w)) for h, w in shapes)
    device = torch.device(device)
    key = (shapes, str(device))
    hit = _pyramids.get(key)
    if hit is None:
        from . import _native
        starts, acc = [], 0
        for h, w in shapes:
            starts.append(acc)
            acc += h * w
        spatial = torch.as_tensor(shapes, dtype=torch.long, device=device)
        level_start = torch.as_tensor(starts, dtype=torch.long, device=device)
        if device.type == "cuda":
            _native.register_host_values(spatial, [v for hw in shapes for v in hw])
            _native.register_host_values(level_start, starts)
        hit = _pyramids[key] = (spatial, level_start)
    return hit


def prepare_data(self, srcs, masks, pos_embeds):
    """Replacement for ``DeformableTransformer.prepare_data`` (deformable_transformer.py:69-94; same arguments, same six results):
    per level the feature map, its padding mask and its positional embedding (+ the level embedding) flattened and concatenated
    over the levels, the valid ratios of every frame -- and the pyramid's ``spatial_shapes`` / ``level_start_index`` taken from
    :func:`interned_pyramid` instead of being pushed to the device again (the reference's ``torch.as_tensor(..., device=...)`` of a
    Python list is a blocking host-to-device copy: it waits for everything queued on the stream)."""
    feats, pads, embeds = [], [], []
    for lvl, (src, mask, pos_embed) in enumerate(zip(srcs, masks, pos_embeds)):
        feats.append(src.flatten(2).transpose(1, 2))
        pads.append(mask.flatten(1))
        embeds.append(pos_embed.flatten(2).transpose(1, 2) + self.level_embed[lvl].view(1, 1, -1))
    src_flatten, mask_flatten, lvl_pos_embed_flatten = torch.cat(feats, 1), torch.cat(pads, 1), torch.cat(embeds, 1)
    spatial_shapes, level_start_index = interned_pyramid([src.shape[-2:] for src in srcs], src_flatten.device)
    valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1)
    return src_flatten, mask_flatten, lvl_pos_embed_flatten, spatial_shapes, level_start_index, valid_ratios


_int_tensors = {}       # (values, dtype, device) -> tensor: interned for good (tens of bytes each)


class _InterningTorch:
    """Stands in for the name ``torch`` INSIDE ``src.models.devis_transformer`` (only there): every attribute is torch's own, except
    that ``torch.tensor(<short list of Python ints>, device=<device>)`` -- how the stacks build each frame's ``temporal_offsets`` on
    every forward (devis_transformer.py:100, 113, 149) -- returns one interned device tensor per (values, device) instead of a new
    blocking host-to-device copy.  The tensors are read-only by convention (the stacks only index with them)."""

    def __init__(self, real):
        self.__dict__["_real"] = real

    def __getattr__(self, name):
        return getattr(self._real, name)

    def tensor(self, data, *args, **kwargs):
        if (not args and isinstance(data, (list, tuple)) and 0 < len(data) <= 256 and all(type(v) is int for v in data) and
                kwargs.get("device") is not None and set(kwargs) <= {"device", "dtype"}):
            device = self._real.device(kwargs["device"])
            key = (tuple(data), kwargs.get("dtype"), str(device))
            hit = _int_tensors.get(key)
            if hit is None:
                hit = _int_tensors[key] = self._real.tensor(data, dtype=kwargs.get("dtype"), device=device)
                if device.type == "cuda":
                    from . import _native
                    _native.register_host_values(hit, data)
            return hit
        return self._real.tensor(data, *args, **kwargs)


def patch_transformer(deformable_transformer_module, devis_transformer_module=None):
    """Opt-in: make the reference's encoder stacks (``DeformableTransformerEncoder`` and its subclass
    ``DeVISTransformerEncoder``) use the cached :func:`get_reference_points`, and ``DeformableTransformer.prepare_data`` (inherited
    by ``DeVISTransformer``) the interning :func:`prepare_data`.  Pass the imported reference module
    ``src.models.deformable_transformer``; with ``src.models.devis_transformer`` as the second argument the temporal offsets its
    stacks build on every forward are interned too (:class:`_InterningTorch`), which takes the last blocking copies out of a
    training step.  Returns the replaced ``get_reference_points`` (to undo the first patch; :func:`unpatch_transformer` undoes
    all of them)."""
    cls = deformable_transformer_module.DeformableTransformerEncoder
    previous = cls.__dict__.get("get_reference_points")
    cls.get_reference_points = staticmethod(get_reference_points)
    top = getattr(deformable_transformer_module, "DeformableTransformer", None)
    if top is not None and "prepare_data" in top.__dict__ and top.__dict__["prepare_data"] is not prepare_data:
        top._devis_amd_prepare_data = top.__dict__["prepare_data"]
        top.prepare_data = prepare_data
    if devis_transformer_module is not None and not isinstance(getattr(devis_transformer_module, "torch", None), _InterningTorch):
        devis_transformer_module.torch = _InterningTorch(devis_transformer_module.torch)
    return previous


def unpatch_transformer(deformable_transformer_module, previous_get_reference_points, devis_transformer_module=None):
    """Undo :func:`patch_transformer` (tests)."""
    deformable_transformer_module.DeformableTransformerEncoder.get_reference_points = previous_get_reference_points
    top = getattr(deformable_transformer_module, "DeformableTransformer", None)
    if top is not None and "_devis_amd_prepare_data" in top.__dict__:
        top.prepare_data = top.__dict__["_devis_amd_prepare_data"]
        del top._devis_amd_prepare_data
    if devis_transformer_module is not None and isinstance(getattr(devis_transformer_module, "torch", None), _InterningTorch):
        devis_transformer_module.torch = devis_transformer_module.torch._real
