"""Cached form of the one call-site argument builder the reference exposes as a function:
``DeformableTransformerEncoder.get_reference_points`` (``src/models/deformable_transformer.py:185-198``), which the
DeVIS encoder stack calls once per forward (``devis_transformer.py:95``).  The reference rebuilds two ``linspace`` s and
a ``meshgrid`` per pyramid level per call -- each ``linspace`` reads its bounds from the device tensor
``spatial_shapes``, i.e. synchronises -- although the centre grids only depend on the pyramid.  Here the grids are
cached per pyramid (read from the host copy ``_native.shapes_hint`` already keeps per ``spatial_shapes`` tensor), and
only the per-call arithmetic with ``valid_ratios`` runs, in the reference's order: the result is bit-identical
(``tests/test_host_cpu.py::test_cached_reference_points_match_the_reference_call_site``, fixture made by the
reference).

Opt-in wiring, no DeVIS source change::

    import src.models.deformable_transformer as dt
    devis_amd.patch_transformer(dt)

The other call-site tensors (temporal offsets, repeated shapes: ``devis_transformer.py:97-118,146-158``) are built
inline in the reference's ``forward`` s; the attention modules consume them as they come
(``TemporalMSDeformAttnBase._frame_table``: one cached, synchronisation-free device table per list of offsets).
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


def patch_transformer(deformable_transformer_module):
    """Opt-in: make the reference's encoder stacks (``DeformableTransformerEncoder`` and its subclass
    ``DeVISTransformerEncoder``) use the cached :func:`get_reference_points`.  Pass the imported reference module
    ``src.models.deformable_transformer``.  Returns the replaced static method (to undo the patch)."""
    cls = deformable_transformer_module.DeformableTransformerEncoder
    previous = cls.__dict__.get("get_reference_points")
    cls.get_reference_points = staticmethod(get_reference_points)
    return previous
