"""GPU: forward+backward time of the DeVIS temporal attention modules (one clip, DeVIS sizes), fused
single-launch path vs the reference's 2*T-call pattern (module.fused = False).  Not the headline bench."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devis_amd.modules import TemporalMSDeformAttnDecoder, TemporalMSDeformAttnEncoder

PYR = {"A": [(45, 80), (23, 40), (12, 20), (6, 10)], "B": [(100, 167), (50, 84), (25, 42), (13, 21)]}
dev = "cuda:0"
T, C = 6, 256


def run(kind, pyr, fused, pad=1, prep=True, reps=10):
    torch.manual_seed(0)
    shapes = torch.tensor(PYR[pyr], device=dev)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    t_shapes = shapes.repeat(T - 1, 1)
    t_lsi = torch.cat((t_shapes.new_zeros(1), t_shapes.prod(1).cumsum(0)[:-1]))
    offs = [torch.tensor([t for t in range(-f, T - f) if t != 0], device=dev) for f in range(T)]
    if kind == "decoder":
        mod = TemporalMSDeformAttnDecoder(T, C, 4, T - 1, 8, 4, 4).to(dev)
        query = torch.randn(1, T * 300, C, device=dev, requires_grad=True)
        ref = torch.rand(1, T * 300, 4, 2, device=dev)
    else:
        mod = TemporalMSDeformAttnEncoder(T, C, 4, T - 1, 8, 4, 4).to(dev)
        query = torch.randn(T, S, C, device=dev, requires_grad=True)
        ref = torch.rand(T, S, 4, 2, device=dev)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    src = torch.randn(T, S, C, device=dev, requires_grad=True)
    mod.fused = fused
    mod.value_pad_heads = pad
    mod.fused_prep = prep

    def step():
        out = mod(query, ref, src, (shapes, t_shapes), (lsi, t_lsi), offs)[0]
        torch.autograd.grad(out.square().sum(), (query, src) + tuple(mod.parameters()))

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{kind:8s} pyramid {pyr} (S={S}) fused={fused!s:5s} value_pad_heads={pad} fused_prep={prep!s:5s}: {ms:8.3f} ms per layer fwd+bwd (whole module incl. Linears/softmax)", flush=True)


for kind, pyr in (("decoder", "A"), ("encoder", "A"), ("decoder", "B"), ("encoder", "B")):
    run(kind, pyr, True, 1, True)
    run(kind, pyr, True, 1, False)
    run(kind, pyr, True, 0, False)
    run(kind, pyr, False, 0, False)
