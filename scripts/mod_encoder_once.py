"""GPU probe: a few fwd+bwd steps of the temporal encoder module (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devis_amd.modules import TemporalMSDeformAttnDecoder, TemporalMSDeformAttnEncoder
PYR = {"A": [(45, 80), (23, 40), (12, 20), (6, 10)], "B": [(100, 167), (50, 84), (25, 42), (13, 21)]}
dev = "cuda:0"; T, C = 6, 256
kind = os.environ.get("KIND", "encoder"); pyr = os.environ.get("PYR", "A")
torch.manual_seed(0)
shapes = torch.tensor(PYR[pyr], device=dev)
lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
S = int(shapes.prod(1).sum())
t_shapes = shapes.repeat(T - 1, 1)
t_lsi = torch.cat((t_shapes.new_zeros(1), t_shapes.prod(1).cumsum(0)[:-1]))
offs = [torch.tensor([t for t in range(-f, T - f) if t != 0], device=dev) for f in range(T)]
if kind == "decoder":
    mod = TemporalMSDeformAttnDecoder(T, C, 4, T - 1, 8, 4, 4).to(dev)
    query = torch.randn(1, T * 300, C, device=dev, requires_grad=True); ref = torch.rand(1, T * 300, 4, 2, device=dev)
else:
    mod = TemporalMSDeformAttnEncoder(T, C, 4, T - 1, 8, 4, 4).to(dev)
    query = torch.randn(T, S, C, device=dev, requires_grad=True); ref = torch.rand(T, S, 4, 2, device=dev)
with torch.no_grad():
    for p in mod.parameters(): p.copy_(torch.randn_like(p) * 0.05)
src = torch.randn(T, S, C, device=dev, requires_grad=True)
for _ in range(5):
    out = mod(query, ref, src, (shapes, t_shapes), (lsi, t_lsi), offs)[0]
    torch.autograd.grad(out.square().sum(), (query, src) + tuple(mod.parameters()))
torch.cuda.synchronize(); print("done")
