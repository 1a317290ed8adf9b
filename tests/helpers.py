"""Shared helpers for the parity tests (test infrastructure; may import the oracle)."""
import numpy as np
import torch

from oracle import msda_oracle as O

PYR_A = [(45, 80), (23, 40), (12, 20), (6, 10)]        # 360x640 input (DeVIS test size), S = 4820
PYR_B = [(100, 167), (50, 84), (25, 42), (13, 21)]     # 800x1333 input, S = 22223


def make_inputs(seed, N, M, D, Lq, shapes, P, loc_mode="wide", dtype=np.float32, value_scale=0.01):
    """Seeded synthetic inputs, rounded ONCE to `dtype` precision (so every path sees identical
    numbers).  loc_mode: 'unit' = rand in [0,1) (reference test.py), 'wide' = rand*1.4-0.2 (exercises
    the out-of-range rules)."""
    rng = np.random.default_rng(seed)
    shapes = np.asarray(shapes, dtype=np.int64)
    L = shapes.shape[0]
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = rng.random((N, S, M, D)) * value_scale
    loc = rng.random((N, Lq, M, L, P, 2))
    if loc_mode == "wide":
        loc = loc * 1.4 - 0.2
    aw = rng.random((N, Lq, M, L, P)) + 1e-5
    aw = aw / aw.sum(axis=(-1, -2), keepdims=True)
    grad_out = rng.standard_normal((N, Lq, M * D))
    return dict(value=value.astype(dtype), shapes=shapes, lsi=O.level_start_index(shapes),
                loc=loc.astype(dtype), aw=aw.astype(dtype), grad_out=grad_out.astype(dtype))


def round_to(arrs, torch_dtype):
    """Round numpy float64 arrays through a torch storage dtype (bf16/f16/f32) and back to float64."""
    out = {}
    for k, v in arrs.items():
        if isinstance(v, np.ndarray) and v.dtype.kind == "f":
            out[k] = torch.from_numpy(np.asarray(v, dtype=np.float64)).to(torch_dtype).double().numpy()
        else:
            out[k] = v
    return out


def oracle_fwd_bwd(d, dtype=np.float64):
    """Oracle outputs (C restatement) for an input dict in the given arithmetic dtype."""
    v, l, a, g = (np.asarray(d[k], dtype=dtype) for k in ("value", "loc", "aw", "grad_out"))
    out = O.forward(v, d["shapes"], d["lsi"], l, a)
    gv, gl, ga = O.backward(v, d["shapes"], d["lsi"], l, a, g)
    return out, gv, gl, ga


def temporal_reference(value, shapes, lsi, ftab, loc_c, aw_c, loc_t, aw_t, grad_out=None):
    """Oracle for the fused temporal op = the reference's call pattern (ms_deform_attn.py:325-364):
    per frame t one call on value[t] and one on the `window` frames ftab[t] stacked along the level
    axis with spatial_shapes.repeat(window).  numpy float64 in/out.  value [T,S,M,D] (one clip)."""
    T, S, M, D = value.shape
    W = ftab.shape[1]
    t_shapes = np.tile(shapes, (W, 1))
    t_lsi = O.level_start_index(t_shapes)
    outs, gv = [], np.zeros_like(value)
    gl_c, ga_c, gl_t, ga_t = (np.zeros_like(x) for x in (loc_c, aw_c, loc_t, aw_t))
    for t in range(T):
        stacked = value[ftab[t]].reshape(1, W * S, M, D)
        o1 = O.forward(value[t][None], shapes, lsi, loc_c[t][None], aw_c[t][None])
        o2 = O.forward(stacked, t_shapes, t_lsi, loc_t[t][None], aw_t[t][None])
        outs.append(o1 + o2)
        if grad_out is not None:
            g = grad_out[t][None]
            a, b, c = O.backward(value[t][None], shapes, lsi, loc_c[t][None], aw_c[t][None], g)
            gv[t] += a[0]; gl_c[t] = b[0]; ga_c[t] = c[0]
            a, b, c = O.backward(stacked, t_shapes, t_lsi, loc_t[t][None], aw_t[t][None], g)
            np.add.at(gv, ftab[t], a[0].reshape(W, S, M, D))      # index_put-add, repeats accumulate
            gl_t[t] = b[0]; ga_t[t] = c[0]
    out = np.concatenate(outs, 0)
    if grad_out is None:
        return out
    return out, gv, gl_c, ga_c, gl_t, ga_t


def make_temporal_inputs(seed, T, W, M, D, Lq, shapes, Pc, Pt, ftab=None, dtype=np.float32):
    rng = np.random.default_rng(seed)
    shapes = np.asarray(shapes, dtype=np.int64)
    L = shapes.shape[0]
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    if ftab is None:   # every other frame, ascending (devis_transformer.py:147-150)
        assert W == T - 1
        ftab = np.array([[f for f in range(T) if f != t] for t in range(T)], dtype=np.int32)
    value = rng.random((T, S, M, D)) * 0.01
    loc_c = rng.random((T, Lq, M, L, Pc, 2)) * 1.4 - 0.2
    loc_t = rng.random((T, Lq, M, W * L, Pt, 2)) * 1.4 - 0.2
    aw = rng.random((T, Lq, M, L * Pc + W * L * Pt)) + 1e-5
    aw = aw / aw.sum(-1, keepdims=True)
    aw_c = aw[..., :L * Pc].reshape(T, Lq, M, L, Pc)
    aw_t = aw[..., L * Pc:].reshape(T, Lq, M, W * L, Pt)
    grad_out = rng.standard_normal((T, Lq, M * D))
    f = lambda x: np.ascontiguousarray(x.astype(dtype))
    return dict(value=f(value), shapes=shapes, lsi=O.level_start_index(shapes), ftab=np.ascontiguousarray(ftab),
                loc_c=f(loc_c), aw_c=f(aw_c), loc_t=f(loc_t), aw_t=f(aw_t), grad_out=f(grad_out))


def pixel_centres(shapes):
    """[S, 2] normalised (x, y) centres of the pixels of a pyramid, in query order (= the encoder's reference points,
    deformable_transformer.py:185-198 with valid_ratios = 1)."""
    out = []
    for h, w in np.asarray(shapes, dtype=np.int64).tolist():
        ys, xs = np.meshgrid((np.arange(h) + 0.5) / h, (np.arange(w) + 0.5) / w, indexing="ij")
        out.append(np.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    return np.concatenate(out, 0)


def localise(loc, shapes, sigma_px, seed, levels_per_slot=None):
    """Replace sampling locations [..., Lq = S, M, LL, P, 2] by encoder-like ones: the query's own pixel centre plus
    N(0, sigma_px^2) pixels of the sampled level (LL = a multiple of the number of levels: slot-major, level-minor)."""
    rng = np.random.default_rng(seed)
    shapes = np.asarray(shapes, dtype=np.int64)
    L = shapes.shape[0]
    c = pixel_centres(shapes)                                           # [S, 2]
    LL = loc.shape[-3]
    wh = np.stack([shapes[:, 1], shapes[:, 0]], -1).astype(np.float64)  # (W, H) per level
    wh = np.tile(wh, (LL // L, 1))                                      # [LL, 2]
    noise = rng.standard_normal(loc.shape) * sigma_px / wh[:, None, :]
    return (c[:, None, None, None, :] + noise).astype(loc.dtype)
