import os, sys, torch
DEV = "cuda:0"
R, C = 28920, 256
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode == "side":
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream()); torch.cuda.set_stream(side)
sx = torch.randn(R, C, device=DEV); sy = torch.randn(R, C, device=DEV)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        a = sx * sx.mean(); b = (sy * 2.0).sum(0)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
share = os.environ.get("DBG_SHARE", "1") == "1"
pool = torch.cuda.graph_pool_handle()
with torch.cuda.graph(ga, pool=pool):
    a = sx * sx.mean()                  # graph A: a small temporary (the mean) and a large output
with torch.cuda.graph(gb, pool=pool if share else None):
    b = (sy * 2.0).sum(0)               # graph B: a SMALL output
for it in range(5):
    x = torch.randn(R, C, device=DEV); y = torch.randn(R, C, device=DEV)
    sx.copy_(x); sy.copy_(y)
    ga.replay()
    if mode == "sync": torch.cuda.synchronize()
    e = torch.randn_like(a); e2 = (a * e).sum()                 # eager kernels between the two launches
    gb.replay()
    torch.cuda.synchronize()
    ok = torch.allclose(b, (y * 2.0).sum(0), rtol=1e-3, atol=1e-2)
    print(mode, "share_pool", share, "iteration", it, "B's small output", "ok" if ok else "WRONG %s" % b[:3].tolist(), flush=True)
