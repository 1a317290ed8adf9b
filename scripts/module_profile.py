"""GPU probe: the kernels of ONE TemporalMSDeformAttnDecoder layer call (one clip, forward + backward incl. its Linears), for
`rocprofv3 --kernel-trace --stats -- python3 scripts/module_profile.py [steps]`: what surrounds the operator at module level."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from devis_amd.modules import TemporalMSDeformAttnDecoder


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    device = torch.device("cuda:0")
    T_, q_, C_ = 6, 300, 256
    gm = torch.Generator(device="cpu").manual_seed(11)
    shp = torch.tensor(bench.PYRAMIDS["A"], dtype=torch.int64, device=device)
    lsi_ = torch.cat((shp.new_zeros(1), shp.prod(1).cumsum(0)[:-1]))
    S_ = int(shp.prod(1).sum())
    tsh = shp.repeat(T_ - 1, 1)
    tlsi = torch.cat((tsh.new_zeros(1), tsh.prod(1).cumsum(0)[:-1]))
    offs = [torch.tensor([t for t in range(-f, T_ - f) if t != 0], device=device) for f in range(T_)]
    mod = TemporalMSDeformAttnDecoder(T_, C_, 4, T_ - 1, 8, 4, 4).to(device)
    with torch.no_grad():
        for prm in mod.parameters():
            prm.copy_(torch.randn(prm.shape, generator=gm).to(device) * 0.05)
    qry = torch.randn(1, T_ * q_, C_, generator=gm).to(device).requires_grad_(True)
    refp = (torch.rand(1, T_ * q_, 4, 2, generator=gm) * 0.8 + 0.1).to(device)
    srcm = torch.randn(T_, S_, C_, generator=gm).to(device).requires_grad_(True)
    wgt = torch.randn(1, T_ * q_, C_, generator=gm).to(device)

    def mstep():
        out = mod(qry, refp, srcm, (shp, tsh), (lsi_, tlsi), offs)[0]
        torch.autograd.grad((out * wgt).sum(), (qry, srcm))
    for _ in range(5):
        mstep()
    torch.cuda.synchronize()
    print("eager ms per step:", round(bench._event_ms(mstep, steps, 0), 4))


if __name__ == "__main__":
    main()
