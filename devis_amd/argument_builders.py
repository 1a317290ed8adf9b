"""Cached builders of the small device tensors the transformer hands to the attention modules on every forward
(SURVEY section 8, row f-4).  The reference rebuilds all of them per call with Python loops and a dozen tiny
kernels each -- ``prepare_data`` (``src/models/deformable_transformer.py:69-94``), ``get_reference_points``
(``:185-198``), the temporal offsets / repeated shapes of ``devis_transformer.py:94-118,147-158`` -- although they
only depend on the clip's pyramid and length.  Here they are built once per key and reused; values are identical
(same formulas, same dtypes).  Nothing here is on the operator's data path; a DeVIS integration calls these
instead of the inline code.
"""
import torch

_level_cache, _grid_cache, _temporal_cache = {}, {}, {}
_MAX_ENTRIES = 32


def _remember(cache, key, value):
    if len(cache) >= _MAX_ENTRIES:
        cache.clear()
    cache[key] = value
    return value


def level_tables(shapes, device):
    """``(spatial_shapes [L,2] int64, level_start_index [L] int64)`` on `device` for the pyramid `shapes` =
    ``[(H_0, W_0), ...]`` (ref ``prepare_data``: ``torch.as_tensor(spatial_shapes, long)``, ``cat(zeros(1),
    prod(1).cumsum(0)[:-1])``)."""
    key = (tuple((int(h), int(w)) for h, w in shapes), str(device))
    hit = _level_cache.get(key)
    if hit is not None:
        return hit
    spatial_shapes = torch.as_tensor(key[0], dtype=torch.long, device=device)
    level_start_index = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
    return _remember(_level_cache, key, (spatial_shapes, level_start_index))


def reference_points(shapes, valid_ratios, device):
    """Encoder reference points ``[N, S, L, 2]`` (ref ``get_reference_points``, ``deformable_transformer.py:185-198``):
    pixel centres of every level, normalised by the valid part of the (padded) frame.  The per-level centre grids
    (two linspaces + a meshgrid per level per call in the reference) are cached per pyramid; the division and
    multiplication by `valid_ratios` [N, L, 2] -- the only per-call arithmetic -- are done in the reference's order,
    so the result is bit-identical."""
    key = (tuple((int(h), int(w)) for h, w in shapes), str(device))
    grids = _grid_cache.get(key)
    if grids is None:
        grids = []
        for H_, W_ in key[0]:
            ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device),
                                          torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device),
                                          indexing='ij')
            grids.append((ref_y.reshape(-1)[None], ref_x.reshape(-1)[None]))
        _remember(_grid_cache, key, grids)
    per_level = []
    for lvl, ((H_, W_), (gy, gx)) in enumerate(zip(key[0], grids)):
        ref_y = gy / (valid_ratios[:, None, lvl, 1] * H_)
        ref_x = gx / (valid_ratios[:, None, lvl, 0] * W_)
        per_level.append(torch.stack((ref_x, ref_y), -1))
    points = torch.cat(per_level, 1)
    return points[:, :, None] * valid_ratios[:, None]


def temporal_tables(spatial_shapes, n_frames, device, t_window=None):
    """``(temporal_offsets, temporal_spatial_shapes, temporal_level_start_index)`` of the temporal encoder / decoder.

    ``t_window=None``: connect-all (decoder always, encoder with ``enc_connect_all_embeddings``):
    ``temporal_offsets[t] = [-t .. T-1-t] \\ {0}`` and the shapes repeated ``T-1`` times
    (ref ``devis_transformer.py:100-103, 147-155``).  Otherwise the encoder's window of ``t_window`` neighbours,
    mirrored at the clip's ends (ref ``:105-115``).  `spatial_shapes` is the device tensor of :func:`level_tables`.
    The offsets come back as the same list of tensor OBJECTS on every call, which is also what lets
    ``TemporalMSDeformAttnBase._frame_table`` reuse its table."""
    key = (tuple(map(tuple, spatial_shapes.tolist())), int(n_frames), t_window, str(device))
    hit = _temporal_cache.get(key)
    if hit is not None:
        return hit
    T_ = int(n_frames)
    offsets = []
    if t_window is None:
        repeated = spatial_shapes.repeat(T_ - 1, 1)
        for curr in range(T_):
            offsets.append(torch.tensor([t for t in range(-curr, T_ - curr) if t != 0], device=device))
    else:
        repeated = spatial_shapes.repeat(t_window, 1)
        neighbours = [t for t in range(-t_window // 2, (t_window // 2) + 1) if t != 0]
        for curr in range(T_):
            offsets.append(torch.tensor([-t if (curr + t < 0 or curr + t > T_ - 1) else t for t in neighbours], device=device))
    start = torch.cat((repeated.new_zeros((1,)), repeated.prod(1).cumsum(0)[:-1]))
    return _remember(_temporal_cache, key, (offsets, repeated, start))
