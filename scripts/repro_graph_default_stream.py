"""GPU: pure-PyTorch reproduction (nothing of this package is imported) of what `devis_amd.graphed` works around.

On this image (PyTorch 2.10 + ROCm 7) a callable made by ``torch.cuda.make_graphed_callables`` and replayed on the LEGACY DEFAULT
stream returns, from the second replay on, a parameter gradient that the backward graph did not write: the tensor still holds
what the forward graph last left in that part of the graph's memory pool (here: the scalar ``mean``).  Outputs and input
gradients are right; on any non-default stream, or with a device synchronisation between the forward replay and the backward,
everything is right.

    python scripts/repro_graph_default_stream.py            # default stream: iteration 0 ok, 1..3 MISMATCH
    python scripts/repro_graph_default_stream.py side       # a side stream: all ok
    python scripts/repro_graph_default_stream.py sync       # default stream + synchronize after the forward replay: all ok

`devis_amd/graphs.py` therefore captures and replays on a stream of its own (tests/test_modules_gpu.py covers that from the
default stream, over two signatures and six replays)."""
import sys

import torch
from torch import nn

DEV = "cuda:0"
R, C = 28920, 256


class Shifted(nn.Module):
    def __init__(self):
        super().__init__()
        self.b = nn.Parameter(torch.zeros(C))

    def forward(self, x, r):
        scale = (r[torch.arange(r.shape[0] - 1, -1, -1, device=r.device)].contiguous() * 1.0).mean()   # small forward temporaries
        return (x + self.b) * scale


def main(mode):
    if mode == "side":
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(side)
    m = Shifted().to(DEV)

    def inputs(seed):
        return (torch.randn(R, C, device=DEV, generator=torch.Generator(DEV).manual_seed(seed)).requires_grad_(True),
                torch.rand(720, 4, 2, device=DEV, generator=torch.Generator(DEV).manual_seed(seed + 1)) * 0.8 + 0.1)

    graphed = torch.cuda.make_graphed_callables(m, tuple(t.detach().clone().requires_grad_(t.requires_grad) for t in inputs(1)))
    bad = 0
    for it in range(4):
        x, r = inputs(10 + it)
        out = graphed(x, r)
        if mode == "sync":
            torch.cuda.synchronize()
        w = torch.randn_like(out)
        gx, gb = torch.autograd.grad((out * w).sum(), [x, m.b])
        torch.cuda.synchronize()
        want = (w * r.mean()).sum(0)
        ok = torch.allclose(gb, want, rtol=1e-3, atol=1e-2)
        bad += 0 if ok else 1
        print("%-7s replay %d: bias gradient %s" % (mode, it, "ok" if ok else "MISMATCH got %s want %s" % (gb[:3].tolist(), want[:3].tolist())),
              flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1] if len(sys.argv) > 1 else "default") else 0)
