"""Rebuilds the module test cases stored in tests/golden/mod_*.npz (made from the REFERENCE modules by
tests/golden/make_golden.py) on our modules, and compares every returned quantity and gradient."""
import numpy as np
import torch

from conftest import golden

C, M, L, T, Q = 32, 4, 2, 3, 5      # dimensions used by make_golden.py


def build(name, device, dtype):
    from devis_amd.modules import MSDeformAttn, TemporalMSDeformAttnDecoder, TemporalMSDeformAttnEncoder
    g = golden(name)
    if name.startswith("mod_plain"):
        mod = MSDeformAttn(d_model=C, n_levels=L, n_heads=M, n_points=3)
    elif name == "mod_temporal_enc":
        mod = TemporalMSDeformAttnEncoder(T, C, L, T - 1, M, 3, 2)
    elif name == "mod_temporal_enc_window":
        mod = TemporalMSDeformAttnEncoder(5, C, L, 2, M, 3, 2)
    else:
        mod = TemporalMSDeformAttnDecoder(T, C, L, T - 1, M, 3, 2,
                                          dec_instance_aware_att="not_instance_aware" not in name)
    state = {k[len("state/"):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("state/")}
    assert sorted(state) == sorted(mod.state_dict())          # reference checkpoints load key for key
    mod = mod.to(dtype=torch.float64)          # before loading: load_state_dict casts to the param dtype
    mod.load_state_dict(state)
    mod = mod.to(device=device, dtype=dtype)
    return mod, g


def _t(g, key, device, dtype):
    x = torch.from_numpy(g[key])
    return x.to(device=device, dtype=dtype if x.is_floating_point() else x.dtype)


def run(name, device, dtype, fused=True):
    mod, g = build(name, device, dtype)
    if hasattr(mod, "fused"):
        mod.fused = fused
    query = _t(g, "in/query", device, dtype).requires_grad_(True)
    src = _t(g, "in/input_flatten", device, dtype).requires_grad_(True)
    ref = _t(g, "in/reference_points", device, dtype)
    if name.startswith("mod_plain"):
        ret = mod(query, ref, src, _t(g, "in/spatial_shapes", device, dtype),
                  _t(g, "in/level_start_index", device, dtype), _t(g, "in/padding_mask", device, dtype))
    else:
        shapes = tuple(_t(g, "in/spatial_shapes/%d" % i, device, dtype) for i in range(2))
        lsi = tuple(_t(g, "in/level_start_index/%d" % i, device, dtype) for i in range(2))
        n = len([k for k in g if k.startswith("in/temporal_offsets/")])
        offs = [_t(g, "in/temporal_offsets/%d" % i, device, dtype) for i in range(n)]
        ret = mod(query, ref, src, shapes, lsi, offs)
    out = ret[0]
    w = _t(g, "loss_weight", device, dtype)
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad((out * w).sum(), [query, src] + list(params.values()))
    got = {"out": out, "grad/query": grads[0], "grad/input_flatten": grads[1]}
    for (k, _), gr in zip(params.items(), grads[2:]):
        got["grad/" + k] = gr
    if "aux/n_frames" in g:
        assert len(ret) == 5                                   # ms_deform_attn.py:414
        n = int(g["aux/n_frames"])
        assert isinstance(ret[1], list) and isinstance(ret[2], list) and len(ret[1]) == n and len(ret[2]) == n
        for i in range(n):
            got["aux/curr_loc/%d" % i] = ret[1][i]
            got["aux/temp_loc/%d" % i] = ret[2][i]
        got["aux/aw_curr"], got["aux/aw_temp"] = ret[3], ret[4]
    else:
        assert len(ret) == 2 and ret[1] is None
    return got, g


def compare(got, g, rtol, atol):
    for k, v in got.items():
        exp = g[k]
        a = v.detach().double().cpu().numpy()
        assert a.shape == exp.shape, (k, a.shape, exp.shape)
        scale = max(1.0, float(np.abs(exp).max()))
        err = float(np.abs(a - exp).max())
        assert err <= atol * scale + rtol * scale, (k, err, scale)
