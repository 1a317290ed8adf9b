#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own Python code.

Runs only in the build container (needs /root/reference, read-only); the GPU box never sees the
reference, only the .npz files this script wrote.  Fixtures hold tensors only (inputs + expected
outputs + expected gradients) -- no reference source.

What is imported from the reference (SURVEY.md Appendix B recipe):
  * ``ms_deform_attn_core_pytorch``  (src/models/ops/functions/ms_deform_attn_func.py:102-122)
    -- the reference's oracle for the operator; gradients come from autograd through it in fp64.
  * ``MSDeformAttn``, ``TemporalMSDeformAttnEncoder``, ``TemporalMSDeformAttnDecoder``
    (src/models/ops/modules/ms_deform_attn.py) with the unbuilt CUDA extension stubbed by a
    function that routes to ``ms_deform_attn_core_pytorch``.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_OPS = "/root/reference/src/models/ops"


def import_reference():
    stub = types.ModuleType("MultiScaleDeformableAttention")   # ms_deform_attn_func.py:18
    sys.modules["MultiScaleDeformableAttention"] = stub
    pkg = types.ModuleType("refops")
    pkg.__path__ = [REF_OPS]
    sys.modules["refops"] = pkg
    F = importlib.import_module("refops.functions")
    Mo = importlib.import_module("refops.modules")
    core = F.ms_deform_attn_core_pytorch

    class _ViaCore(torch.autograd.Function):
        """Lets the reference nn.Modules run on CPU: forward = the reference oracle, backward =
        autograd through the reference oracle (so module-level gradients are reference-made)."""

        @staticmethod
        def forward(ctx, value, shapes, lsi, loc, aw, step):
            ctx.save_for_backward(value, shapes, loc, aw)
            return core(value, shapes, loc, aw)

        @staticmethod
        def backward(ctx, go):
            value, shapes, loc, aw = ctx.saved_tensors
            with torch.enable_grad():
                v, l, a = (t.detach().requires_grad_(True) for t in (value, loc, aw))
                out = core(v, shapes, l, a)
                gv, gl, ga = torch.autograd.grad(out, (v, l, a), go)
            return gv, None, None, gl, ga, None

    # the modules call MSDeformAttnFunction.apply(...) -> swap the class the module file bound
    Mo.ms_deform_attn.MSDeformAttnFunction = _ViaCore
    return F, Mo


def lsi_of(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def op_case(core, name, value, shapes, loc, aw, seed, extra=None):
    """Run the reference oracle in fp64 (+autograd) and in fp32 (forward), save everything."""
    g = torch.Generator().manual_seed(seed + 1000)
    # inputs are rounded to float32 ONCE, before any path sees them, and stored as float32: the
    # fp64 and the fp32 checks then run on bit-identical inputs (and the files are half the size)
    value, loc, aw = (t.float() for t in (value, loc, aw))
    v, l, a = (t.double().detach().requires_grad_(True) for t in (value, loc, aw))
    out = core(v, shapes, l, a)
    grad_out = torch.randn(out.shape, generator=g, dtype=torch.float64)
    gv, gl, ga = torch.autograd.grad(out, (v, l, a), grad_out)
    out32 = core(value, shapes, loc, aw)
    # Points sitting EXACTLY on h_im == -1 or w_im == -1: the reference CUDA kernels skip them
    # (cuh:288, strict '>'), so every gradient of such a point is 0 there, whereas autograd through
    # grid_sample returns the one-sided derivative for grad_loc.  Output and the other gradients
    # agree (the in-range corner has weight exactly 0).  We follow the CUDA kernel -- the thing being
    # replaced -- so the golden grad_sampling_loc is zeroed on this measure-zero set and the mask kept.
    wh = torch.stack([shapes[:, 1], shapes[:, 0]], -1).double()[None, None, None, :, None, :]
    on_edge = ((loc.double() * wh - 0.5) == -1).any(-1)
    gl = gl * (~on_edge)[..., None]
    d = dict(value=value.numpy(), spatial_shapes=shapes.numpy(),
             level_start_index=lsi_of(shapes).numpy(), sampling_locations=loc.numpy(),
             attention_weights=aw.numpy(), grad_output=grad_out.numpy(),
             out=out.detach().numpy(), grad_value=gv.numpy(), grad_sampling_loc=gl.numpy(),
             grad_attn_weight=ga.numpy(), out_f32=out32.numpy(), seed=np.int64(seed),
             on_minus_one_edge=on_edge.numpy(),
             torch_version=np.array(torch.__version__))
    if extra:
        d.update(extra)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print("wrote %-34s out %s  |out|max %.3e" % (name, tuple(out.shape), out.abs().max().item()))


def rand_inputs(seed, N, M, D, Lq, shapes, P, loc_mode="unit"):
    g = torch.Generator().manual_seed(seed)
    L = shapes.shape[0]
    S = int(shapes.prod(1).sum())
    value = torch.rand(N, S, M, D, generator=g, dtype=torch.float64) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g, dtype=torch.float64)
    if loc_mode == "wide":
        loc = loc * 1.4 - 0.2
    aw = torch.rand(N, Lq, M, L, P, generator=g, dtype=torch.float64) + 1e-5
    aw = aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
    return value, loc, aw


def make_op_fixtures(F):
    core = F.ms_deform_attn_core_pytorch
    small = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)

    # 1. the reference test.py procedure: same shapes, same seed, same draw order (test.py:19-35)
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    torch.manual_seed(3)
    value = torch.rand(N, 30, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    aw = torch.rand(N, Lq, M, L, P) + 1e-5
    aw /= aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
    op_case(core, "op_testpy_shape", value, small, loc, aw, seed=3)

    # 2. out-of-range locations + hand-placed boundary points (cuh:288 rule, cuh:56-78 corners).
    # Power-of-two maps so that the boundary pixel coordinates are exact in fp32 AND fp64.
    pow2 = torch.as_tensor([(8, 4), (4, 2)], dtype=torch.long)
    value, loc, aw = rand_inputs(11, 1, 2, 4, 12, pow2, 4, "wide")
    H, W = 8.0, 4.0
    pix = [-1.0, -0.75, -0.5, 0.0, 0.25, H - 1, H - 0.5, H - 0.25, H, -1.25, H + 0.5]
    for i, py in enumerate(pix):          # query i, head 0, level 0: point 0 has y on a boundary,
        loc[0, i, 0, 0, 0, 1] = (py + 0.5) / H
        px = min(py, W + 0.5)             # point 1 has x on a boundary
        loc[0, i, 0, 0, 1, 0] = (px + 0.5) / W
    op_case(core, "op_out_of_range", value, pow2, loc, aw, seed=11)

    # 3. batch > 1 (im2col_step chunking: ms_deform_attn_cuda.cu:50-75)
    value, loc, aw = rand_inputs(12, 6, 2, 4, 5, small, 3, "wide")
    op_case(core, "op_batched_im2col", value, small, loc, aw, seed=12)

    # 4. generic head dims (test.py:61-84 picks these to hit each backward kernel variant)
    for D in (30, 32, 64, 71, 1025):
        value, loc, aw = rand_inputs(20 + D, 1, 2, D, 3, small, 2, "wide")
        op_case(core, "op_generic_D%d" % D, value, small, loc, aw, seed=20 + D)

    # 5. BASELINE.json configs[0]: L=1 64x64, Nq=100, M=8, K=4, C=256
    one = torch.as_tensor([(64, 64)], dtype=torch.long)
    value, loc, aw = rand_inputs(31, 1, 8, 32, 100, one, 4, "wide")
    op_case(core, "op_cfg1", value, one, loc, aw, seed=31)

    # 6. many levels (the temporal call stacks (T-1)*L levels): 5 x a tiny pyramid
    pyr = torch.as_tensor([(6, 4), (3, 2), (2, 2), (1, 1)] * 5, dtype=torch.long)
    value, loc, aw = rand_inputs(32, 1, 8, 32, 7, pyr, 4, "wide")
    op_case(core, "op_many_levels", value, pyr, loc, aw, seed=32)

    # 7. DeVIS-shaped but small: M=8, D=32, L=4, P=4, two batch rows, odd query count
    pyr = torch.as_tensor([(12, 20), (6, 10), (3, 5), (2, 3)], dtype=torch.long)
    value, loc, aw = rand_inputs(33, 2, 8, 32, 37, pyr, 4, "wide")
    op_case(core, "op_devis_small", value, pyr, loc, aw, seed=33)


def randomise(module, g):
    """_reset_parameters zeroes the offset/attention Linear weights (ms_deform_attn.py:65,77-78),
    which would make the output independent of the query -- draw every parameter instead."""
    module.double()      # the bias built from float32 thetas (ms_deform_attn.py:66,74) stays f32 otherwise
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=g, dtype=p.dtype) * 0.3)


def module_case(name, module, args, arg_names, grad_names, seed, n_aux=0):
    g = torch.Generator().manual_seed(seed + 500)
    leaves = {k: a for k, a in zip(arg_names, args) if k in grad_names}
    for a in leaves.values():
        a.requires_grad_(True)
    ret = module(*args)
    out = ret[0]
    w = torch.randn(out.shape, generator=g, dtype=out.dtype)
    params = dict(module.named_parameters())
    grads = torch.autograd.grad((out * w).sum(), list(leaves.values()) + list(params.values()))
    d = {"out": out.detach().numpy(), "loss_weight": w.numpy(), "seed": np.int64(seed)}
    for k, v in module.state_dict().items():
        d["state/" + k] = v.detach().numpy()
    for (k, _), gr in zip(list(leaves.items()) + list(params.items()), grads):
        d["grad/" + k] = gr.numpy()
    for k, a in zip(arg_names, args):
        if isinstance(a, torch.Tensor):
            d["in/" + k] = a.detach().numpy()
        elif a is None:
            pass
        elif isinstance(a, (tuple, list)):
            for i, x in enumerate(a):
                d["in/%s/%d" % (k, i)] = x.detach().numpy()
    if n_aux:                      # the decoder's 5-tuple (ms_deform_attn.py:414)
        cur_locs, tmp_locs, aw_c, aw_t = ret[1:]
        d["aux/n_frames"] = np.int64(len(cur_locs))
        for i, x in enumerate(cur_locs):
            d["aux/curr_loc/%d" % i] = x.detach().numpy()
        for i, x in enumerate(tmp_locs):
            d["aux/temp_loc/%d" % i] = x.detach().numpy()
        d["aux/aw_curr"] = aw_c.detach().numpy()
        d["aux/aw_temp"] = aw_t.detach().numpy()
    else:
        assert ret[1] is None
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print("wrote %-34s out %s  |out|max %.3e" % (name, tuple(out.shape), out.abs().max().item()))


def make_module_fixtures(Mo):
    torch.set_default_dtype(torch.float64)
    C, M, L, q, T = 32, 4, 2, 5, 3
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = 30
    lsi = lsi_of(shapes)

    # ---- plain MSDeformAttn (ms_deform_attn.py:30-132): 2-d and 4-d reference points
    for ref_dim in (2, 4):
        g = torch.Generator().manual_seed(40 + ref_dim)
        mod = Mo.MSDeformAttn(d_model=C, n_levels=L, n_heads=M, n_points=3)
        randomise(mod, g)
        N = 2
        query = torch.randn(N, q, C, generator=g)
        ref = torch.rand(N, q, L, ref_dim, generator=g)
        src = torch.randn(N, S, C, generator=g)
        mask = torch.rand(N, S, generator=g) < 0.2
        module_case("mod_plain_ref%d" % ref_dim, mod, [query, ref, src, shapes, lsi, mask],
                    ["query", "reference_points", "input_flatten", "spatial_shapes",
                     "level_start_index", "padding_mask"], ("query", "input_flatten"), 40 + ref_dim)

    # freshly initialised parameters (pins _reset_parameters, ms_deform_attn.py:64-82,169-213)
    torch.manual_seed(0)
    fresh = {}
    for nm, mod in (("plain", Mo.MSDeformAttn(C, L, M, 3)),
                    ("temporal", Mo.TemporalMSDeformAttnEncoder(T, C, L, T - 1, M, 3, 2))):
        for k, v in mod.state_dict().items():
            if "value_proj.weight" in k or "output_proj.weight" in k:
                continue            # xavier draws: RNG-dependent, not a contract
            fresh[nm + "/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "mod_fresh_init.npz"), **fresh)
    print("wrote mod_fresh_init")

    temporal_shapes = shapes.repeat(T - 1, 1)
    t_lsi = lsi_of(temporal_shapes)
    offsets = [torch.tensor([t for t in range(-f, T - f) if t != 0]) for f in range(T)]
    tnames = ["query", "reference_points", "input_flatten", "spatial_shapes", "level_start_index",
              "temporal_offsets"]

    # ---- temporal encoder (ms_deform_attn.py:417-464): Lq = S, 2-d reference only
    g = torch.Generator().manual_seed(50)
    enc = Mo.TemporalMSDeformAttnEncoder(T, C, L, T - 1, M, 3, 2)
    randomise(enc, g)
    query = torch.randn(T, S, C, generator=g)
    ref = torch.rand(T, S, L, 2, generator=g)
    src = torch.randn(T, S, C, generator=g)
    module_case("mod_temporal_enc", enc, [query, ref, src, (shapes, temporal_shapes), (lsi, t_lsi), offsets],
                tnames, ("query", "input_flatten"), 50)

    # ---- temporal encoder, windowed connection with mirrored (repeated) frames
    # (devis_transformer.py:103-113 builds offsets like [+1,+1] at the clip borders)
    g = torch.Generator().manual_seed(51)
    T5, win = 5, 2
    enc_w = Mo.TemporalMSDeformAttnEncoder(T5, C, L, win, M, 3, 2)
    randomise(enc_w, g)
    w_offsets = []
    for f in range(T5):
        fo = []
        for tf in (-1, 1):
            fo.append(-tf if (f + tf < 0 or f + tf > T5 - 1) else tf)
        w_offsets.append(torch.tensor(fo))
    w_shapes = shapes.repeat(win, 1)
    query = torch.randn(T5, S, C, generator=g)
    ref = torch.rand(T5, S, L, 2, generator=g)
    src = torch.randn(T5, S, C, generator=g)
    module_case("mod_temporal_enc_window", enc_w,
                [query, ref, src, (shapes, w_shapes), (lsi, lsi_of(w_shapes)), w_offsets],
                tnames, ("query", "input_flatten"), 51)

    # ---- temporal decoder (ms_deform_attn.py:288-414): 2-d / 4-d refs, instance-aware on/off
    for name, ref_dim, aware, seed in (("mod_temporal_dec_ref2", 2, True, 60),
                                       ("mod_temporal_dec_ref4", 4, True, 61),
                                       ("mod_temporal_dec_not_instance_aware", 2, False, 62),
                                       ("mod_temporal_dec_ref4_not_instance_aware", 4, False, 63)):
        g = torch.Generator().manual_seed(seed)
        dec = Mo.TemporalMSDeformAttnDecoder(T, C, L, T - 1, M, 3, 2, dec_instance_aware_att=aware)
        randomise(dec, g)
        query = torch.randn(1, T * q, C, generator=g)
        ref = torch.rand(1, T * q, L, ref_dim, generator=g)
        src = torch.randn(T, S, C, generator=g)
        module_case(name, dec, [query, ref, src, (shapes, temporal_shapes), (lsi, t_lsi), offsets],
                    tnames, ("query", "input_flatten"), seed, n_aux=4)
    torch.set_default_dtype(torch.float32)


def make_config_sized_fixtures(Mo):
    """SURVEY 8c harness row: one config-sized case per temporal module (DeVIS cfg3 dimensions: C=256, M=8, L=4, T=6,
    300 queries per frame / Lq = S, pyramid of the 360x640 test size), run through the REFERENCE modules in fp64 on the
    CPU.  Parameters and inputs come from seeds (tests/module_cases.py::cfg_build, shared with the test), the fixture
    stores statistics and a strided subsample of every output and gradient: a few hundred KB."""
    sys.path.insert(0, os.path.dirname(HERE))
    import module_cases
    for kind, cls in (("dec", Mo.TemporalMSDeformAttnDecoder), ("enc", Mo.TemporalMSDeformAttnEncoder)):
        got = module_cases.cfg_run(kind, cls)
        d = {}
        for k, v in got.items():
            smp = module_cases.cfg_sample(v)
            d[k + "/sample"], d[k + "/stats"] = smp["sample"], smp["stats"]
        np.savez_compressed(os.path.join(HERE, "cfg_%s.npz" % kind), **d)
        print("wrote cfg_%s: %d arrays, |out| max %.3e" % (kind, len(got), np.abs(got["out"]).max()))


def import_reference_transformer():
    """The reference's call-site code: src/models/deformable_transformer.py and devis_transformer.py
    imported as a package whose __init__ is skipped (it pulls in torchvision backbones) and whose only other import
    outside ops/ -- util.misc.inverse_sigmoid, not used by anything recorded here -- is a placeholder that raises."""
    import_reference()
    top = types.ModuleType("refsrc"); top.__path__ = ["/root/reference/src"]; sys.modules["refsrc"] = top
    pkg = types.ModuleType("refsrc.models"); pkg.__path__ = ["/root/reference/src/models"]; sys.modules["refsrc.models"] = pkg
    util = types.ModuleType("refsrc.util"); util.__path__ = []; sys.modules["refsrc.util"] = util
    misc = types.ModuleType("refsrc.util.misc")

    def inverse_sigmoid(*a, **k):
        raise NotImplementedError("placeholder: not part of the recorded call sites")
    misc.inverse_sigmoid = inverse_sigmoid
    sys.modules["refsrc.util.misc"] = misc
    return (importlib.import_module("refsrc.models.deformable_transformer"),
            importlib.import_module("refsrc.models.devis_transformer"))


class _Recorder(torch.nn.Module):
    """Stands in for an encoder / decoder LAYER: records the arguments the reference's stack hands it."""

    def __init__(self):
        super().__init__()
        self.calls = []

    def forward(self, *args, **kwargs):
        self.calls.append((args, kwargs))
        return args[0]


def make_call_site_fixtures():
    """What the reference's transformer stacks hand the attention modules (devis_transformer.py:94-121,146-169;
    deformable_transformer.py:185-198): encoder reference points, temporal offsets (connect-all and windowed, clip ends
    mirrored), repeated shapes / level starts.  `frames[t]` = arange(T)[offsets[t] + t], i.e. the frames the
    reference's `value[temporal_offsets[t] + t]` (ops/modules/ms_deform_attn.py:339,445) selects."""
    DT, DV = import_reference_transformer()
    g = torch.Generator().manual_seed(77)
    shapes = [(12, 20), (6, 10), (3, 5)]
    ss = torch.as_tensor(shapes, dtype=torch.long)
    lsi = lsi_of(ss)
    S = int(ss.prod(1).sum())
    d = {"spatial_shapes": ss.numpy(), "level_start_index": lsi.numpy()}
    for T in (5, 6):
        vr = torch.rand(T, len(shapes), 2, generator=g) * 0.4 + 0.6
        d["T%d/valid_ratios" % T] = vr.numpy()
        d["T%d/reference_points" % T] = DT.DeformableTransformerEncoder.get_reference_points(ss, vr, torch.device("cpu")).numpy()
        src = torch.zeros(T, S, 8)
        for mode, connect_all, window in (("all", True, 2), ("win2", False, 2), ("win4", False, 4)):
            enc = DV.DeVISTransformerEncoder(_Recorder(), 1, window, connect_all)
            enc(src, ss, lsi, vr)
            (a, kw), = enc.layers[0].calls          # (_get_clones deep-copied the recorder)
            offsets = kw["temporal_offsets"]
            key = "T%d/enc_%s/" % (T, mode)
            d[key + "offsets"] = torch.stack(offsets).numpy()
            d[key + "frames"] = torch.stack([torch.arange(T)[o + t] for t, o in enumerate(offsets)]).numpy()
            d[key + "temporal_shapes"] = a[3][1].numpy()
            d[key + "temporal_lsi"] = a[4][1].numpy()
            assert torch.equal(a[2], torch.from_numpy(d["T%d/reference_points" % T]))
        dec = DV.DeVISTransformerDecoder(_Recorder(), 1)
        dec.refine_reference_point = lambda lid, out, ref, inter, inter_ref: (ref, inter + [out], inter_ref + [ref])
        dec(torch.zeros(1, 7 * T, 8), torch.rand(1, 7 * T, 2, generator=g), src, ss, lsi, vr)
        (a, kw), = dec.layers[0].calls
        offsets = kw["temporal_offsets"]
        key = "T%d/dec/" % T
        d[key + "offsets"] = torch.stack(offsets).numpy()
        d[key + "frames"] = torch.stack([torch.arange(T)[o + t] for t, o in enumerate(offsets)]).numpy()
        d[key + "temporal_shapes"] = a[4][1].numpy()
        d[key + "temporal_lsi"] = a[5][1].numpy()
    np.savez_compressed(os.path.join(HERE, "args_call_sites.npz"), **d)
    print("wrote args_call_sites: %d arrays" % len(d))


if __name__ == "__main__":
    if not os.path.isdir(REF_OPS):
        sys.exit("reference not present: golden vectors can only be regenerated in the build container")
    F, Mo = import_reference()
    if "--call-sites-only" in sys.argv:
        make_call_site_fixtures()
        sys.exit(0)
    if "--config-sized-only" not in sys.argv:
        make_op_fixtures(F)
        make_module_fixtures(Mo)
    make_config_sized_fixtures(Mo)
    make_call_site_fixtures()
    os.system("du -sh %s" % HERE)
