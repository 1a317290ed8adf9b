import os, sys, torch
from torch import nn
DEV = "cuda:0"
R, C = 28920, 256
class Shifted(nn.Module):
    def __init__(self):
        super().__init__(); self.b = nn.Parameter(torch.zeros(C))
    def forward(self, x, r):
        v = os.environ.get("DBG_FWD", "gather")
        if v == "gather": scale = (r[torch.arange(r.shape[0] - 1, -1, -1, device=r.device)].contiguous() * 1.0).mean()
        elif v == "mean": scale = r.mean()
        elif v == "mulmean": scale = (r * 1.0).mean()
        elif v == "flipmean": scale = (r.flip(0).contiguous() * 1.0).mean()
        elif v == "sum": scale = (r * 1.0).sum() / r.numel()
        elif v == "elem": scale = (r * 1.0)[3, 1, 0]
        return (x + self.b) * scale
m = Shifted().to(DEV)
def inputs(seed):
    return (torch.randn(R, C, device=DEV, generator=torch.Generator(DEV).manual_seed(seed)).requires_grad_(True),
            torch.rand(720, 4, 2, device=DEV, generator=torch.Generator(DEV).manual_seed(seed + 1)) * 0.8 + 0.1)
sx, sr = (t.detach().clone().requires_grad_(t.requires_grad) for t in inputs(1))
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        o = m(sx, sr); torch.autograd.grad(o, [sx, m.b], torch.empty_like(o))
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
fg, bg = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
pool = torch.cuda.graph_pool_handle()
with torch.cuda.graph(fg, pool=pool):
    so = m(sx, sr)
sgo = torch.empty_like(so)
keepalive = []
with torch.cuda.graph(bg, pool=None if os.environ.get("DBG_SEPPOOL") else pool):
    if os.environ.get("DBG_PADBWD"):
        keepalive = [torch.empty(n, device=DEV) for n in (128, 256, 512, 1024, 2048, 4096)]      # take the freed forward blocks first
    if os.environ.get("DBG_MANUALBWD"):
        scale_saved = so.grad_fn.next_functions[1][0] if False else None
        gy = sgo * (sr[torch.arange(sr.shape[0] - 1, -1, -1, device=DEV)].contiguous() * 1.0).mean()
        gb = gy.sum(0); gx = gy
    else:
        gx, gb = torch.autograd.grad(so, [sx, m.b], sgo)
prev_want = None
bad = 0
for it in range(4):
    x, r = inputs(10 + it)
    sx.detach().copy_(x.detach()); sr.copy_(r)
    fg.replay()
    w = torch.randn_like(so); sgo.copy_(w)
    bg.replay()
    torch.cuda.synchronize()
    want = (w * r.mean()).sum(0)
    ok = torch.allclose(gb, want, rtol=1e-3, atol=1e-2)
    okx = torch.allclose(gx, w * r.mean(), rtol=1e-3, atol=1e-3)
    print("   input gradient", "ok" if okx else "WRONG", end=" | ")
    if os.environ.get("DBG_BG2"):
        bg.replay(); torch.cuda.synchronize()
        print("   after a second backward replay:", "ok" if torch.allclose(gb, want, rtol=1e-3, atol=1e-2) else "still wrong %s" % gb[:3].tolist())
    stale = prev_want is not None and torch.allclose(gb, prev_want, rtol=1e-3, atol=1e-2)
    print("manual two-graph replay", it, "ok" if ok else "MISMATCH %s vs %s%s" % (gb[:3].tolist(), want[:3].tolist(), "  (= the PREVIOUS replay's result)" if stale else ""), flush=True)
    prev_want = want
