"""GPU tests of the nn.Modules (HIP kernels underneath) against the golden fixtures made from the
REFERENCE modules: identical state_dict + inputs -> outputs, the decoder's auxiliary returns and all
gradients.  fp64 runs through the generic kernels, fp32 through the tile kernels (D = 8 -> 2 lanes/row)."""
import pytest
import torch

import module_cases
from conftest import golden_names

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MODULE_FIXTURES = [n for n in golden_names("mod_") if n != "mod_fresh_init"]


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", MODULE_FIXTURES)
def test_modules_fp64(name, fused):
    if name.startswith("mod_plain") and not fused:
        pytest.skip("plain module has a single call pattern")
    got, g = module_cases.run(name, "cuda:0", torch.float64, fused=fused)
    module_cases.compare(got, g, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("name", MODULE_FIXTURES)
def test_modules_fp32(name):
    got, g = module_cases.run(name, "cuda:0", torch.float32, fused=True)
    module_cases.compare(got, g, rtol=1e-4, atol=1e-4)      # BASELINE bar for fp32


def test_devis_sized_decoder_layer_runs_and_fused_equals_loop():
    """T=6, 300 queries/frame, C=256, M=8, L=4, K=4 on the 360x640 pyramid: fused == 2*T-call loop."""
    from devis_amd.modules import TemporalMSDeformAttnDecoder
    from helpers import PYR_A
    torch.manual_seed(0)
    T, q, C = 6, 300, 256
    dev = "cuda:0"
    mod = TemporalMSDeformAttnDecoder(T, C, 4, T - 1, 8, 4, 4).to(dev)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    shapes = torch.tensor(PYR_A, device=dev)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    t_shapes = shapes.repeat(T - 1, 1)
    t_lsi = torch.cat((t_shapes.new_zeros(1), t_shapes.prod(1).cumsum(0)[:-1]))
    offs = [torch.tensor([t for t in range(-f, T - f) if t != 0], device=dev) for f in range(T)]
    query = torch.randn(1, T * q, C, device=dev, requires_grad=True)
    ref = torch.rand(1, T * q, 4, 2, device=dev)
    src = torch.randn(T, S, C, device=dev, requires_grad=True)
    outs = []
    for fused in (True, False):
        mod.fused = fused
        ret = mod(query, ref, src, (shapes, t_shapes), (lsi, t_lsi), offs)
        assert ret[0].shape == (1, T * q, C) and len(ret[1]) == T and ret[2][0].shape == (1, q, 8, 20, 4, 2)
        gq, gs = torch.autograd.grad(ret[0].square().sum(), (query, src))
        outs.append((ret[0].detach(), gq, gs))
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= 1e-3 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-6)])
@pytest.mark.parametrize("d", [2, 4])
@pytest.mark.parametrize("W", [0, 3])
def test_prep_function_matches_torch_ops(dtype, tol, d, W):
    """MSDeformPrepFunction (SURVEY f-2: joint softmax + sampling locations in one fused pass) against the
    torch ops of ref ms_deform_attn.py:112-121 / :252-258 -- outputs and every gradient, 2-d and box
    reference points, with and without a temporal part."""
    import torch.nn.functional as F
    from devis_amd.functions import MSDeformPrepFunction
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(11 + d + W)
    R, M, L, Pc, Pt = 37, 8, 3, 4, 2
    shapes = torch.tensor([[9, 7], [5, 4], [3, 2]], device=DEV)
    mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).to(DEV, dtype).requires_grad_(True)
    off_c, lg_c = mk(R, M, L, Pc, 2), mk(R, M, L * Pc)
    ref_c = torch.rand(R, L, d, generator=g, dtype=torch.float64).to(DEV, dtype).requires_grad_(True)
    off_t = lg_t = ref_t = None
    if W:
        off_t, lg_t = mk(R, M, W * L, Pt, 2), mk(R, M, W * L * Pt)
        ref_t = torch.rand(R, W * L, d, generator=g, dtype=torch.float64).to(DEV, dtype).requires_grad_(True)

    def torch_path():
        norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).to(dtype)
        def loc(ref, off, nrm, P):
            r = ref[:, None, :, None, :]
            return r + off / nrm[None, None, :, None, :] if d == 2 else r[..., :2] + off / P * r[..., 2:] * 0.5
        logits = lg_c if not W else torch.cat([lg_c, lg_t], 2)
        w = F.softmax(logits, -1)
        outs = [loc(ref_c, off_c, norm, Pc), w[..., :L * Pc].reshape(R, M, L, Pc)]
        if W:
            outs += [loc(ref_t, off_t, norm.repeat(W, 1), Pt), w[..., L * Pc:].reshape(R, M, W * L, Pt)]
        return outs

    def fused_path():
        lc, lt, ac, at = MSDeformPrepFunction.apply(off_c, off_t, lg_c, lg_t, ref_c, ref_t, shapes)
        return [lc, ac] + ([lt, at] if W else [])

    leaves = [t for t in (off_c, lg_c, ref_c, off_t, lg_t, ref_t) if t is not None]
    results = []
    for path in (torch_path, fused_path):
        outs = path()
        gen = torch.Generator().manual_seed(5)
        cot = [torch.randn(o.shape, generator=gen, dtype=torch.float64).to(DEV, dtype) for o in outs]
        grads = torch.autograd.grad(outs, leaves, cot)
        results.append([o.detach() for o in outs] + list(grads))
    for a, b in zip(*results):
        assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("d", [2, 4])
@pytest.mark.parametrize("W", [0, 3])
def test_prep_fused_function_equals_prep_function(d, W):
    """MSDeformPrepFusedFunction (offsets / logits as column slices of the single query-side GEMM's output, one
    gradient matrix back) against MSDeformPrepFunction on the same numbers: outputs and all gradients."""
    from devis_amd.functions import MSDeformPrepFunction, MSDeformPrepFusedFunction
    DEV, dtype = "cuda:0", torch.float64
    g = torch.Generator().manual_seed(23 + d + W)
    R, M, L, Pc, Pt = 29, 8, 3, 4, 2
    shapes = torch.tensor([[9, 7], [5, 4], [3, 2]], device=DEV)
    cols = MSDeformPrepFusedFunction._cols(M, L, W, Pc, Pt if W else 1)
    width = cols[3][1] if W else cols[2][1] + 0
    if not W:
        cols = MSDeformPrepFusedFunction._cols(M, L, 0, Pc, 1)
        width = cols[3][1]
    y = torch.randn(R, width, generator=g, dtype=torch.float64).to(DEV, dtype).requires_grad_(True)
    ref_c = torch.rand(R, L, d, generator=g, dtype=torch.float64).to(DEV, dtype).requires_grad_(True)
    ref_t = torch.rand(R, W * L, d, generator=g, dtype=torch.float64).to(DEV, dtype).requires_grad_(True) if W else None

    def unfused():
        off_c = y[:, cols[0][0]:cols[0][1]].reshape(R, M, L, Pc, 2)
        lg_c = y[:, cols[2][0]:cols[2][1]].reshape(R, M, L * Pc)
        off_t = y[:, cols[1][0]:cols[1][1]].reshape(R, M, W * L, Pt, 2) if W else None
        lg_t = y[:, cols[3][0]:cols[3][1]].reshape(R, M, W * L * Pt) if W else None
        return MSDeformPrepFunction.apply(off_c, off_t, lg_c, lg_t, ref_c, ref_t, shapes)

    def fused():
        return MSDeformPrepFusedFunction.apply(y, ref_c, ref_t, shapes, M, L, W, Pc, Pt if W else 1)

    leaves = [t for t in (y, ref_c, ref_t) if t is not None]
    results = []
    for path in (unfused, fused):
        outs = [o for o in path() if o is not None]
        gen = torch.Generator().manual_seed(5)
        cot = [torch.randn(o.shape, generator=gen, dtype=torch.float64).to(DEV, dtype) for o in outs]
        grads = torch.autograd.grad(outs, leaves, cot)
        results.append([o.detach() for o in outs] + list(grads))
    for a, b in zip(*results):
        assert a.shape == b.shape
        assert (a - b).abs().max().item() <= 1e-12 * max(1.0, b.abs().max().item())


def test_temporal_decoder_module_is_graph_capturable():
    """The whole TemporalMSDeformAttnDecoder call (value_proj, fused query-side GEMM, pre-op pass, frame table, fused
    operator, output_proj; forward AND backward) contains no host synchronisation and no allocation outside torch's
    allocator, so torch.cuda.make_graphed_callables can capture it (the per-layer call DeVIS issues is
    host-bound when run eagerly); the graphed module reproduces the eager one on new inputs."""
    from devis_amd.modules import TemporalMSDeformAttnDecoder
    torch.manual_seed(0)
    T, q, C, M, L = 6, 300, 256, 8, 4
    shapes = torch.tensor(module_cases.CFG["pyramid"], dtype=torch.long, device=DEV)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    t_shapes = shapes.repeat(T - 1, 1)
    t_lsi = torch.cat((t_shapes.new_zeros((1,)), t_shapes.prod(1).cumsum(0)[:-1]))
    offsets = [torch.tensor([t for t in range(-f, T - f) if t != 0], device=DEV) for f in range(T)]
    mod = TemporalMSDeformAttnDecoder(T, C, L, T - 1, M, 4, 4).to(DEV)
    with torch.no_grad():
        for p in mod.parameters():
            p.normal_(0, 0.05)

    def call(query, ref, src):
        return mod(query, ref, src, (shapes, t_shapes), (lsi, t_lsi), offsets)[0]

    def inputs(seed):
        g = torch.Generator(device="cpu").manual_seed(seed)
        mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
        return (mk(1, T * q, C).requires_grad_(True), (torch.rand(1, T * q, L, 2, generator=g) * 0.8 + 0.1).to(DEV),
                mk(T, S, C).requires_grad_(True))

    graphed = torch.cuda.make_graphed_callables(call, inputs(1))
    for seed in (2, 3):
        a, b = inputs(seed), inputs(seed)
        out_g = graphed(*a)
        out_e = call(*b)
        w = torch.randn_like(out_e)
        gg = torch.autograd.grad((out_g * w).sum(), (a[0], a[2]))
        ge = torch.autograd.grad((out_e * w).sum(), (b[0], b[2]))
        torch.cuda.synchronize()
        assert torch.allclose(out_g, out_e, rtol=1e-5, atol=1e-6)
        for x, y in zip(gg, ge):
            assert torch.allclose(x, y, rtol=1e-4, atol=1e-5 * float(y.abs().max()))


def test_graphed_helper_reproduces_the_eager_layer_with_parameter_gradients_and_caches_per_shape():
    """``devis_amd.graphed``: the layer with the module's own call signature, forward and backward replayed from
    a HIP graph -- outputs, input gradients AND parameter gradients of the eager module; one graph per signature (a second
    query count captures a second graph, a repeated one does not); the decoder's auxiliary returns come through."""
    import devis_amd
    from devis_amd.modules import TemporalMSDeformAttnDecoder
    torch.manual_seed(0)
    T, C, M, L = 6, 256, 8, 4
    shapes = torch.tensor(module_cases.CFG["pyramid"], dtype=torch.long, device=DEV)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    t_shapes = shapes.repeat(T - 1, 1)
    t_lsi = torch.cat((t_shapes.new_zeros((1,)), t_shapes.prod(1).cumsum(0)[:-1]))
    offsets = [torch.tensor([t for t in range(-f, T - f) if t != 0], device=DEV) for f in range(T)]
    mod = TemporalMSDeformAttnDecoder(T, C, L, T - 1, M, 4, 4).to(DEV)
    with torch.no_grad():
        for p in mod.parameters():
            p.normal_(0, 0.05)

    def inputs(seed, q):
        g = torch.Generator(device="cpu").manual_seed(seed)
        mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
        return (mk(1, T * q, C).requires_grad_(True), (torch.rand(1, T * q, L, 2, generator=g) * 0.8 + 0.1).to(DEV),
                mk(T, S, C).requires_grad_(True), (shapes, t_shapes), (lsi, t_lsi), offsets)

    layer = devis_amd.graphed(mod, inputs(1, 60))
    assert layer.graphs == 1
    params = [p for p in mod.parameters()]

    def check(seed, q, eager_first):
        a, b = inputs(seed, q), inputs(seed, q)
        res_g = layer(*a)
        res_e = mod(*b)
        assert len(res_g) == 5 and len(res_g[1]) == T and res_g[3].shape == res_e[3].shape
        assert res_g[0].requires_grad and not res_g[3].requires_grad and not res_g[1][0].requires_grad      # aux outputs: detached
        w = torch.randn_like(res_e[0])
        if eager_first:
            ge = torch.autograd.grad((res_e[0] * w).sum(), [b[0], b[2]] + params)
        gg = torch.autograd.grad((res_g[0] * w).sum(), [a[0], a[2]] + params)
        if not eager_first:
            ge = torch.autograd.grad((res_e[0] * w).sum(), [b[0], b[2]] + params)
        torch.cuda.synchronize()
        assert torch.allclose(res_g[0], res_e[0], rtol=1e-5, atol=1e-6)
        assert torch.allclose(res_g[3], res_e[3], rtol=1e-5, atol=1e-7) and torch.allclose(res_g[1][2], res_e[1][2], rtol=1e-5, atol=1e-6)
        for (name, _), x, y in zip([("query", 0), ("src", 0)] + list(mod.named_parameters()), gg, ge):
            assert torch.allclose(x, y, rtol=1e-4, atol=2e-5 * max(1e-6, float(y.abs().max()))), (seed, q, name)

    # Training steps belong on a non-default stream (devis_amd.graph_stream): on this PyTorch-ROCm build a backward graph replayed
    # next to default-stream work hands back unwritten bias / weight gradients from the second replay on
    # (scripts/repro_graph_default_stream.py, pure torch).  Each graph is replayed three times, the eager twin's backward now
    # before and now after the graphed one (the orders that exposed it).
    with devis_amd.graph_stream():
        assert torch.cuda.current_stream() != torch.cuda.default_stream()
        for seed, q in ((2, 60), (3, 180), (4, 60), (5, 180), (6, 60), (7, 180)):
            check(seed, q, eager_first=bool(seed % 2))
    assert layer.graphs == 2 and layer.eager_calls == 0     # 60 and 180 queries per frame; the later calls replayed them
    # ... and a training call made ON the default stream runs the module eagerly (one warning), right all the same
    assert torch.cuda.current_stream() == torch.cuda.default_stream()
    with pytest.warns(UserWarning, match="graph_stream"):
        check(8, 60, eager_first=False)
    check(9, 180, eager_first=True)
    assert layer.graphs == 2 and layer.eager_calls == 2
    # inference: first use from inside torch.no_grad() (tracker.py:320-323) -- its own signature (nothing requires grad); replays
    # from the default stream too, several times over, with eager work in between
    with torch.no_grad():
        for seed in (10, 11, 12, 13):
            a = inputs(seed, 60)
            a = tuple(x.detach() if isinstance(x, torch.Tensor) else x for x in a)
            res_g = layer(*a)
            noise = (torch.randn_like(res_g[0]) * res_g[0]).sum()
            res_e = mod(*a)
            torch.cuda.synchronize()
            assert torch.allclose(res_g[0], res_e[0], rtol=1e-5, atol=1e-6) and not res_g[0].requires_grad
            assert torch.allclose(res_g[3], res_e[3], rtol=1e-5, atol=1e-7) and torch.allclose(res_g[1][2], res_e[1][2], rtol=1e-5, atol=1e-6)
    assert layer.graphs == 3 and layer.eager_calls == 2
    # ... and an inference graph follows the parameters: an in-place update (an optimizer step) between two replays
    with torch.no_grad():
        for p in mod.parameters():
            p.mul_(1.25)
        a = tuple(x.detach() if isinstance(x, torch.Tensor) else x for x in inputs(14, 60))
        res_g, res_e = layer(*a), mod(*a)
        torch.cuda.synchronize()
        assert layer.graphs == 3 and torch.allclose(res_g[0], res_e[0], rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError, match="no CPU path"):
        devis_amd.graphed(mod, tuple(x.cpu() if isinstance(x, torch.Tensor) else x for x in inputs(1, 60)))


def test_inference_graph_follows_in_place_parameter_updates():
    """A forward captured under ``torch.no_grad()`` (where the modules cache the concatenated query-side Linear parameters between
    calls) must not bake that cached copy in by address: after an in-place parameter update the replay equals the eager call."""
    from devis_amd.modules import MSDeformAttn
    torch.manual_seed(1)
    C, M, L, N, Lq = 256, 8, 4, 2, 50
    shapes = torch.tensor(module_cases.CFG["pyramid"], dtype=torch.long, device=DEV)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    mod = MSDeformAttn(C, L, M, 4).to(DEV)
    q, ref, src = torch.randn(N, Lq, C, device=DEV), torch.rand(N, Lq, L, 2, device=DEV), torch.randn(N, S, C, device=DEV)
    with torch.no_grad():
        for p in mod.parameters():
            p.normal_(0, 0.05)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                mod(q, ref, src, shapes, lsi, None)             # warm-up: fills the parameter cache, the shapes hint, the BLAS handles
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            out = mod(q, ref, src, shapes, lsi, None)[0]
        for scale in (1.0, 1.5, 0.5):
            for p in mod.parameters():
                p.mul_(scale)
            q.normal_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.allclose(out, mod(q, ref, src, shapes, lsi, None)[0], rtol=1e-5, atol=1e-6), scale


@pytest.mark.parametrize("ac", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_modules_run_under_autocast_in_16_bit_storage(ac):
    """fp32 modules under ``torch.autocast`` (something the reference's extension cannot do: its kernels take ONE scalar type
    for value and locations): ``value_proj`` hands the operator a 16-bit ``value`` like any autocast Linear would, the sampling
    locations stay float32, outputs come back in the autocast dtype and parameters get float32 gradients.  Checked against the
    fp32 run: outputs and every gradient that does not go through a sampling POSITION (those of the offset Linears and of the
    query jump when a 16-bit offset lands in the neighbouring pixel cell of a random feature map: not compared)."""
    from devis_amd.functions import project_value
    from devis_amd.modules import MSDeformAttn, TemporalMSDeformAttnDecoder
    T, C, M, L, q = 6, 256, 8, 4, 60
    shapes = torch.tensor(module_cases.CFG["pyramid"], dtype=torch.long, device=DEV)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    t_shapes = shapes.repeat(T - 1, 1)
    t_lsi = torch.cat((t_shapes.new_zeros((1,)), t_shapes.prod(1).cumsum(0)[:-1]))
    offsets = [torch.tensor([t for t in range(-f, T - f) if t != 0], device=DEV) for f in range(T)]
    gen = torch.Generator().manual_seed(5)
    mk = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    torch.manual_seed(0)
    dec, plain = TemporalMSDeformAttnDecoder(T, C, L, T - 1, M, 4, 4).to(DEV), MSDeformAttn(C, L, M, 4).to(DEV)
    with torch.no_grad():
        for p in list(dec.parameters()) + list(plain.parameters()):
            p.normal_(0, 0.05)
    ref = (torch.rand(1, T * q, L, 2, generator=gen) * 0.8 + 0.1).to(DEV)
    mask = (torch.rand(T, S, generator=gen) < 0.05).to(DEV)
    cases = [(dec, (mk(1, T * q, C).requires_grad_(True), ref, mk(T, S, C).requires_grad_(True), (shapes, t_shapes), (lsi, t_lsi), offsets)),
             (plain, (mk(T, q, C).requires_grad_(True), ref.view(T, q, L, 2), mk(T, S, C).requires_grad_(True), shapes, lsi, mask))]
    for mod, args in cases:
        w = mk(*mod(*args)[0].shape)
        wanted = [(n, p) for n, p in mod.named_parameters() if "sampling_offsets" not in n]
        leaves = [args[2]] + [p for _, p in wanted]

        def run(enabled):
            with torch.autocast("cuda", dtype=ac, enabled=enabled):
                out = mod(*args)[0]
                value = project_value(args[2], mod.value_proj, M, None, 1)
            return out, value.dtype, torch.autograd.grad((out.float() * w).sum(), leaves)
        o32, vdt32, g32 = run(False)
        o16, vdt16, g16 = run(True)
        assert (vdt32, vdt16) == (torch.float32, ac) and o16.dtype == ac and all(g.dtype == torch.float32 for g in g16)
        tol = 1e-2 if ac == torch.bfloat16 else 2e-3
        assert float((o16.detach().float() - o32.detach()).abs().max()) <= tol * float(o32.detach().abs().max())
        for (name, _), a, b in zip([("src", None)] + wanted, g16, g32):
            assert float((a - b).abs().max()) <= 4 * tol * float(b.abs().max()), (type(mod).__name__, name)


@pytest.mark.parametrize("kind", ["plain", "temporal_encoder"])
def test_graphed_helper_on_the_other_layer_types(kind):
    """``devis_amd.graphed`` around ``MSDeformAttn`` (with a padding mask: a bound, non-floating argument) and around
    ``TemporalMSDeformAttnEncoder`` (single-tensor output): four replays under ``graph_stream``, outputs, input and parameter
    gradients against the eager module."""
    import devis_amd
    from devis_amd.modules import MSDeformAttn, TemporalMSDeformAttnEncoder
    torch.manual_seed(3)
    C, M, L = 256, 8, 4
    pyr = [(20, 30), (10, 15), (5, 8), (3, 4)]
    shapes = torch.tensor(pyr, dtype=torch.long, device=DEV)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    gen = torch.Generator(device="cpu")
    mk = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    if kind == "plain":
        N, Lq = 3, 77
        mod = MSDeformAttn(C, L, M, 4).to(DEV)
        mask = (torch.rand(N, S, generator=gen) < 0.1).to(DEV)

        def inputs(seed):
            gen.manual_seed(seed)
            return (mk(N, Lq, C).requires_grad_(True), (torch.rand(N, Lq, L, 2, generator=gen) * 0.8 + 0.1).to(DEV),
                    mk(N, S, C).requires_grad_(True), shapes, lsi, mask)
    else:
        T = 4
        mod = TemporalMSDeformAttnEncoder(T, C, L, T - 1, M, 4, 2).to(DEV)
        t_shapes = shapes.repeat(T - 1, 1)
        t_lsi = torch.cat((t_shapes.new_zeros((1,)), t_shapes.prod(1).cumsum(0)[:-1]))
        offsets = [torch.tensor([t for t in range(-f, T - f) if t != 0], device=DEV) for f in range(T)]

        def inputs(seed):
            gen.manual_seed(seed)
            return (mk(T, S, C).requires_grad_(True), (torch.rand(T, S, L, 2, generator=gen) * 0.8 + 0.1).to(DEV),
                    mk(T, S, C).requires_grad_(True), (shapes, t_shapes), (lsi, t_lsi), offsets)
    with torch.no_grad():
        for p in mod.parameters():
            p.normal_(0, 0.05)
    params = list(mod.parameters())
    first = lambda o: o[0] if isinstance(o, tuple) else o
    with devis_amd.graph_stream():
        layer = devis_amd.graphed(mod, inputs(1))
        for seed in (2, 3, 4, 5):
            a, b = inputs(seed), inputs(seed)
            out_g, out_e = first(layer(*a)), first(mod(*b))
            w = torch.randn_like(out_e)
            if seed % 2:
                ge = torch.autograd.grad((out_e * w).sum(), [b[0], b[2]] + params)
            gg = torch.autograd.grad((out_g * w).sum(), [a[0], a[2]] + params)
            if not seed % 2:
                ge = torch.autograd.grad((out_e * w).sum(), [b[0], b[2]] + params)
            torch.cuda.synchronize()
            assert torch.allclose(out_g, out_e, rtol=1e-5, atol=1e-6)
            for (name, _), x, y in zip([("query", 0), ("src", 0)] + list(mod.named_parameters()), gg, ge):
                assert torch.allclose(x, y, rtol=1e-4, atol=2e-5 * max(1e-6, float(y.abs().max()))), (kind, seed, name)
    assert layer.graphs == 1 and layer.eager_calls == 0


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 8e-3), (torch.float16, 2e-3)], ids=["f32", "bf16", "f16"])
def test_value_proj_gradients_at_clip_size_through_the_split_k_product(dtype, tol):
    """`project_value` at the size of one DeVIS clip (T x S = 28 920 rows, 256 -> 256): its backward computes the weight gradient as
    a batched product over row slices (`_split_k_wgrad`, round 4); all three gradients against the fp64 products of the same rounded
    inputs; the slices' partial sums are added in float32 and rounded once (round 5: one storage-type rounding, not ~28)."""
    from devis_amd.functions import project_value
    gen = torch.Generator().manual_seed(17)
    T, S, C, M = 6, 4820, 256, 8
    lin = torch.nn.Linear(C, C)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(C, C, generator=gen) * 0.05)
        lin.bias.copy_(torch.randn(C, generator=gen) * 0.05)
    x = torch.randn(T, S, C, generator=gen)
    g = torch.randn(T, S, M, C // M, generator=gen)
    lin_d = lin.to("cuda:0", dtype)
    x_d = x.to("cuda:0", dtype).requires_grad_(True)
    g_d = g.to("cuda:0", dtype)
    out = project_value(x_d, lin_d, M, None, 1)
    gx, gw, gb = torch.autograd.grad(out, (x_d, lin_d.weight, lin_d.bias), g_d)
    x64, g64 = x_d.detach().double().cpu().reshape(-1, C), g_d.double().cpu().reshape(-1, C)
    w64 = lin_d.weight.detach().double().cpu()
    want = (g64 @ w64, g64.t() @ x64, g64.sum(0))
    for got, ref in zip((gx.reshape(-1, C), gw, gb), want):
        err = float((got.double().cpu() - ref).abs().max())
        assert err <= tol * max(1.0, float(ref.abs().max())), (err, float(ref.abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_split_k_weight_gradient_on_16_bit_inputs_is_as_accurate_as_one_gemm_on_the_gpu(dtype):
    """the split-K weight gradient takes its partial products in float32 (`bmm(..., out_dtype=float32)`) and rounds
    once -- its error against the fp64 product of the same rounded inputs is the single GEMM's, not that of ~28 rounded partials."""
    from devis_amd.functions import ms_deform_attn_func as F
    gen = torch.Generator().manual_seed(23)
    R = 28920
    g = torch.randn(R, 256, generator=gen).to("cuda:0", dtype)
    x = torch.randn(R, 256, generator=gen).to("cuda:0", dtype)
    exact = g.double().t() @ x.double()
    rms = lambda w: float(((w.double() - exact) ** 2).mean().sqrt() / (exact ** 2).mean().sqrt())
    single, split = rms(g.t() @ x), rms(F._split_k_wgrad(g, x))
    assert F._split_k_wgrad(g, x).dtype == dtype
    assert split <= single * 1.05, (split, single, F._BMM_F32_OUT)


def test_reference_shaped_stack_replays_one_graph_per_layer_without_host_synchronisation():
    """SURVEY 8 row f-4 as written: a stack that builds its call-site tensors the way the reference does --
    ``spatial_shapes`` / ``level_start_index`` from the feature maps' Python sizes on every forward
    (deformable_transformer.py:69-94), each frame's ``temporal_offsets`` with ``torch.tensor(list, device=...)``, ``repeat``-ed
    temporal shapes and their start indices on every forward (devis_transformer.py:146-158) -- driven for three training steps
    with ``devis_amd.patch_transformer(dt, dvt)`` + ``devis_amd.graphed`` layers under
    ``torch.cuda.set_sync_debug_mode("error")``: no host synchronisation, ONE graph per layer, and the same outputs and
    gradients as the eager layers.  (Unpatched, the stand-in's own ``torch.as_tensor(..., device=...)`` -- the reference's line
    87 -- already raises under that mode.)"""
    import types
    import devis_amd
    from devis_amd import _native
    from devis_amd.modules import TemporalMSDeformAttnDecoder, TemporalMSDeformAttnEncoder
    torch.manual_seed(7)
    T, C, M, L, Lq = 4, 256, 8, 4, 50
    pyr = [(24, 40), (12, 20), (6, 10), (3, 5)]
    S = sum(h * w for h, w in pyr)

    class Encoder(torch.nn.Module):                       # the method the reference's encoder stacks look up on their class
        @staticmethod
        def get_reference_points(spatial_shapes, valid_ratios, device):
            raise AssertionError("patched away")

    class Stack(torch.nn.Module):
        """What the replacement needs of DeformableTransformer (deformable_transformer.py:17-68): level_embed, get_valid_ratio,
        and a prepare_data of the reference's shape -- everything from Python sizes, new device tensors on every call."""

        def __init__(self):
            super().__init__()
            self.level_embed = torch.nn.Parameter(torch.randn(L, C) * 0.02)

        def get_valid_ratio(self, mask):
            _, H, W = mask.shape
            return torch.stack([torch.sum(~mask[:, 0, :], 1).float() / W, torch.sum(~mask[:, :, 0], 1).float() / H], -1)

        def prepare_data(self, srcs, masks, pos_embeds):
            flat = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
            spatial_shapes = torch.as_tensor([(s.shape[2], s.shape[3]) for s in srcs], dtype=torch.long, device=flat.device)
            level_start_index = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
            return flat, None, None, spatial_shapes, level_start_index, None

    # a module namespace of its own, as src.models.devis_transformer is one: its `torch` is what patch_transformer replaces
    dvt = types.ModuleType("stand_in_devis_transformer")
    exec("import torch\n\n\n"
         "def build_temporal_arguments(spatial_shapes, T_, device):\n"
         "    temporal_offsets = [torch.tensor([t for t in range(-f, T_ - f) if t != 0], device=device) for f in range(T_)]\n"
         "    temporal_spatial_shapes = spatial_shapes.repeat(T_ - 1, 1)\n"
         "    temporal_start = torch.cat((temporal_spatial_shapes.new_zeros((1,)), temporal_spatial_shapes.prod(1).cumsum(0)[:-1]))\n"
         "    return temporal_offsets, temporal_spatial_shapes, temporal_start\n", dvt.__dict__)
    ns = types.SimpleNamespace(DeformableTransformerEncoder=Encoder, DeformableTransformer=Stack)
    stack = Stack().to(DEV)
    enc = TemporalMSDeformAttnEncoder(T, C, L, T - 1, M, 4, 2).to(DEV)
    dec = TemporalMSDeformAttnDecoder(T, C, L, T - 1, M, 4, 2).to(DEV)
    with torch.no_grad():
        for p in list(enc.parameters()) + list(dec.parameters()):
            p.normal_(0, 0.05)
    gen = torch.Generator(device="cpu")
    steps = []          # inputs of all steps made up front (host-side random numbers + their copies are not what is being tested)
    for seed in range(4):
        gen.manual_seed(100 + seed)
        srcs = [torch.randn(T, C, h, w, generator=gen).to(DEV) for h, w in pyr]
        masks = [torch.zeros(T, h, w, dtype=torch.bool, device=DEV) for h, w in pyr]
        q_enc = torch.randn(T, S, C, generator=gen).to(DEV).requires_grad_(True)
        q_dec = torch.randn(1, T * Lq, C, generator=gen).to(DEV).requires_grad_(True)       # the decoder's layout: frames x queries flattened
        ref_enc = (torch.rand(T, S, L, 2, generator=gen) * 0.8 + 0.1).to(DEV)
        ref_dec = (torch.rand(1, T * Lq, L, 2, generator=gen) * 0.8 + 0.1).to(DEV)
        w_enc, w_dec = torch.randn(T, S, C, generator=gen).to(DEV), torch.randn(1, T * Lq, C, generator=gen).to(DEV)
        steps.append((srcs, masks, q_enc, q_dec, ref_enc, ref_dec, w_enc, w_dec))
    torch.cuda.synchronize()

    def one_step(step, enc_layer, dec_layer):
        srcs, masks, q_enc, q_dec, ref_enc, ref_dec, w_enc, w_dec = step
        flat, _, _, spatial_shapes, level_start_index, _ = stack.prepare_data(srcs, masks, srcs)
        flat = flat.detach().requires_grad_(True)
        temporal_offsets, t_shapes, t_lsi = dvt.build_temporal_arguments(spatial_shapes, T, flat.device)
        memory = enc_layer(q_enc, ref_enc, flat, (spatial_shapes, t_shapes), (level_start_index, t_lsi), temporal_offsets)
        memory = memory[0] if isinstance(memory, tuple) else memory
        out = dec_layer(q_dec, ref_dec, memory, (spatial_shapes, t_shapes), (level_start_index, t_lsi), temporal_offsets)[0]
        leaves = [q_enc, q_dec, flat] + list(enc.parameters()) + list(dec.parameters())
        grads = torch.autograd.grad((out * w_dec).sum() + (memory * w_enc).sum(), leaves)
        return out, grads, spatial_shapes, temporal_offsets

    # unpatched: the stand-in's own as_tensor (the reference's line 87) is a synchronising operation
    torch.cuda.set_sync_debug_mode("error")
    try:
        with pytest.raises(RuntimeError, match="synchronizing"):
            stack.prepare_data(steps[0][0], steps[0][1], steps[0][0])
    finally:
        torch.cuda.set_sync_debug_mode("default")

    previous = devis_amd.patch_transformer(ns, dvt)
    try:
        g_enc, g_dec = devis_amd.graphed(enc), devis_amd.graphed(dec)
        with devis_amd.graph_stream():
            one_step(steps[0], g_enc, g_dec)                # interns the pyramid and the offsets, captures both layers
            torch.cuda.synchronize()
            seen, results = [], []
            torch.cuda.set_sync_debug_mode("error")
            try:
                for step in steps[1:]:
                    out, grads, shapes_t, offs_t = one_step(step, g_enc, g_dec)
                    results.append((out.detach().clone(), [g.clone() for g in grads]))      # (a graph's outputs are static buffers: the next replay overwrites them)
                    seen.append((shapes_t, offs_t))
            finally:
                torch.cuda.set_sync_debug_mode("default")
            torch.cuda.synchronize()
            assert g_enc.graphs == 1 and g_dec.graphs == 1 and g_enc.eager_calls == 0 and g_dec.eager_calls == 0
            assert all(s is seen[0][0] for s, _ in seen)                                          # the interned pyramid ...
            assert all(a is b for _, offs in seen for a, b in zip(offs, seen[0][1]))              # ... and offsets
            assert _native.known_host_values(seen[0][0]) == tuple(v for hw in pyr for v in hw)
            for step, (out_g, grads_g) in zip(steps[1:], results):
                out_e, grads_e, _, _ = one_step(step, enc, dec)
                torch.cuda.synchronize()
                assert torch.allclose(out_g, out_e, rtol=1e-5, atol=1e-6)
                for x, y in zip(grads_g, grads_e):
                    assert torch.allclose(x, y, rtol=1e-4, atol=2e-5 * max(1e-6, float(y.abs().max())))
        # without the second argument the offsets are new tensors every step: still one graph (they are graph inputs, copied in)
        devis_amd.argument_builders.unpatch_transformer(ns, previous, dvt)
        previous = devis_amd.patch_transformer(ns)
        with devis_amd.graph_stream():
            for step in steps[:3]:
                out_g, grads_g, _, _ = one_step(step, g_enc, g_dec)
                out_e, grads_e, _, _ = one_step(step, enc, dec)
                torch.cuda.synchronize()
                assert torch.allclose(out_g, out_e, rtol=1e-5, atol=1e-6)
        assert g_enc.graphs == 1 and g_dec.graphs == 1
    finally:
        devis_amd.argument_builders.unpatch_transformer(ns, previous, dvt)
    assert Stack.__dict__["prepare_data"].__qualname__.endswith("Stack.prepare_data") and dvt.torch is torch
