"""GPU tests of the nn.Modules (HIP kernels underneath) against the golden fixtures made from the
REFERENCE modules: identical state_dict + inputs -> outputs, the decoder's auxiliary returns and all
gradients.  fp64 runs through the generic kernels, fp32 through the tile kernels (D = 8 -> 2 lanes/row)."""
import pytest
import torch

import module_cases
from conftest import golden_names

pytestmark = pytest.mark.gpu
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
