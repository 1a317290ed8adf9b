import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import module_cases
import devis_amd
from devis_amd.modules import TemporalMSDeformAttnDecoder
DEV = "cuda:0"
T, C, M, L = 6, 256, 8, 4
shapes = torch.tensor(module_cases.CFG["pyramid"], dtype=torch.long, device=DEV)
lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
S = int(shapes.prod(1).sum())
t_shapes = shapes.repeat(T - 1, 1)
t_lsi = torch.cat((t_shapes.new_zeros((1,)), t_shapes.prod(1).cumsum(0)[:-1]))

def first():
    torch.manual_seed(0)
    q = 300
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
        return (mk(1, T * q, C).requires_grad_(True), (torch.rand(1, T * q, L, 2, generator=g) * 0.8 + 0.1).to(DEV), mk(T, S, C).requires_grad_(True))
    graphed = torch.cuda.make_graphed_callables(call, inputs(1))
    for seed in (2, 3):
        a, b = inputs(seed), inputs(seed)
        out_g = graphed(*a); out_e = call(*b)
        w = torch.randn_like(out_e)
        gg = torch.autograd.grad((out_g * w).sum(), (a[0], a[2])); ge = torch.autograd.grad((out_e * w).sum(), (b[0], b[2]))
        torch.cuda.synchronize()
        print("first", seed, torch.allclose(out_g, out_e, rtol=1e-5, atol=1e-6))

def second():
    torch.manual_seed(0)
    from devis_amd.modules import ms_deform_attn as MM
    if os.environ.get("DBG_PAD0"): MM.TemporalMSDeformAttnBase.value_pad_heads = 0
    if os.environ.get("DBG_NOPREP"): MM.TemporalMSDeformAttnBase.fused_prep = False
    if os.environ.get("DBG_NOFUSED"): MM.TemporalMSDeformAttnBase.fused = False
    offsets = [torch.tensor([t for t in range(-f, T - f) if t != 0], device=DEV) for f in range(T)]
    mod = TemporalMSDeformAttnDecoder(T, C, L, T - 1, M, 4, 4, dec_instance_aware_att=not os.environ.get("DBG_NOINST")).to(DEV)
    with torch.no_grad():
        for p in mod.parameters():
            p.normal_(0, 0.05)
    def inputs(seed, q):
        g = torch.Generator(device="cpu").manual_seed(seed)
        mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
        return (mk(1, T * q, C).requires_grad_(True), (torch.rand(1, T * q, L, 2, generator=g) * 0.8 + 0.1).to(DEV),
                mk(T, S, C).requires_grad_(True), (shapes, t_shapes), (lsi, t_lsi), offsets)
    if os.environ.get("DBG_DIRECT"):
        class W(torch.nn.Module):
            def __init__(self, inner):
                super().__init__(); self.inner = inner
            def forward(self, q_, r_, s_):
                o = self.inner(q_, r_, s_, (shapes, t_shapes), (lsi, t_lsi), offsets)
                return o[0]
        wmod = W(mod)
        i1 = inputs(1, 60)
        with torch.no_grad():
            mod(*i1)
        gm = torch.cuda.make_graphed_callables(wmod, tuple(t.detach().clone().requires_grad_(t.requires_grad) for t in i1[:3]))
        class Lyr:
            def __call__(self, *a):
                return (gm(*a[:3]), None, None, mod(*a)[3].detach(), None) if False else (gm(*a[:3]),)
        layer = Lyr()
    else:
        layer = devis_amd.graphed(mod, inputs(1, 60), aux_grad=bool(os.environ.get("DBG_AUXGRAD")))
    names = ["query", "src"] + [n for n, _ in mod.named_parameters()]
    params = [p for p in mod.parameters()]
    seq = ((2, 60), (3, 60), (4, 60), (5, 60), (6, 60)) if os.environ.get("DBG_ONESIG") else ((2, 60), (3, 180), (4, 60), (5, 180), (6, 60))
    for seed, q in seq:
        a, b = inputs(seed, q), inputs(seed, q)
        res_g = layer(*a); res_e = mod(*b)
        w = torch.randn_like(res_e[0])
        mode = os.environ.get("DBG_MODE", "a")
        if mode == "b":
            ge = torch.autograd.grad((res_e[0] * w).sum(), [b[0], b[2]] + params)
            torch.cuda.synchronize()
        if os.environ.get("DBG_FILL") and "keepx" in globals():
            torch.cuda.synchronize(); keepx.fill_(123.0); torch.cuda.synchronize()
        gg = torch.autograd.grad((res_g[0] * w).sum(), [a[0], a[2]] + params)
        torch.cuda.synchronize()
        globals()["keepx"] = gg[names.index("value_proj.bias")]
        snap = [t.clone() for t in gg]
        torch.cuda.synchronize()
        if mode != "b":
            ge = torch.autograd.grad((res_e[0] * w).sum(), [b[0], b[2]] + params)
            torch.cuda.synchronize()
        i_vb = names.index("value_proj.bias")
        print("   value_proj.bias: snapshot-right-after-graph-backward ok?", torch.allclose(snap[i_vb], ge[i_vb], rtol=1e-4, atol=1e-3),
              "| live tensor after eager backward ok?", torch.allclose(gg[i_vb], ge[i_vb], rtol=1e-4, atol=1e-3), "| x[:4]", gg[i_vb][:4].tolist())
        flat = a[1].flatten()
        v0 = float(gg[i_vb][0])
        hit = (flat == v0).nonzero()
        print("   x[0] found in graphed-call reference_points at", hit.flatten().tolist()[:4], "| in eager-call reference_points at", (b[1].flatten() == v0).nonzero().flatten().tolist()[:4])
        print("second", seed, q, "out", torch.allclose(res_g[0], res_e[0], rtol=1e-5, atol=1e-6), flush=True)
        if os.environ.get("DBG_SNAP") and seed == 2:
            ptr = gg[names.index("value_proj.bias")].data_ptr()
            for seg in torch.cuda.memory_snapshot():
                if seg["address"] <= ptr < seg["address"] + seg["total_size"]:
                    print("   segment", hex(seg["address"]), seg["total_size"], "pool", seg.get("segment_pool_id"), seg.get("segment_type"), "stream", seg.get("stream"))
                    addr = seg["address"]
                    for blk in seg["blocks"]:
                        if addr < seg["address"] + 0x6000:
                            print("      block +%#x size %d state %s" % (addr - seg["address"], blk["size"], blk["state"]))
                        addr += blk["size"]
        for n, x, y in zip(names, gg, ge):
            ok = torch.allclose(x, y, rtol=1e-4, atol=2e-5 * max(1e-6, float(y.abs().max())))
            if not ok:
                print("   MISMATCH", n, tuple(x.shape), "x[min,max]", float(x.min()), float(x.max()), "y[min,max]", float(y.min()), float(y.max()),
                      "ptr", hex(x.data_ptr()), "ref ptr", hex(a[1].data_ptr()), hex(b[1].data_ptr()), "maxdiff", float((x - y).abs().max()))

if os.environ.get("DBG_SIDE"):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(side)
if "1" in sys.argv[1:]:
    first()
second()
