import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import make_inputs, oracle_fwd_bwd
from devis_amd.functions import MSDeformAttnFunction
DEV = "cuda:0"
seed = int(sys.argv[1])
rng = np.random.default_rng(50000 + seed)
route_i = int(rng.integers(0, 8)); lay = int(rng.integers(0, 3)); big = rng.random() < 0.25
D = int(rng.choice([4, 8, 16, 32, 32, 32, 64, 128, 12])); M = int(rng.choice([1, 2, 4, 8, 8, 16]))
L, P = int(rng.integers(1, 6)), int(rng.integers(1, 7))
N, Lq = int(rng.integers(1, 5)), int(rng.integers(1, 3000 if big else 80))
hi = (40, 60) if big else (14, 17)
shapes = [(int(rng.integers(1, hi[0])), int(rng.integers(1, hi[1]))) for _ in range(L)]
mode = "wide" if rng.random() < 0.7 else "unit"
print(dict(D=D, M=M, L=L, P=P, N=N, Lq=Lq, shapes=shapes, mode=mode))
d = make_inputs(seed, N, M, D, Lq, shapes, P, mode, np.float32, value_scale=1.0)
ref = oracle_fwd_bwd(d, np.float32); ref64 = oracle_fwd_bwd(d, np.float64)
f = lambda k: torch.from_numpy(np.asarray(d[k], dtype=np.float64)).to(DEV, torch.float32)
v = f("value").requires_grad_(True); l = f("loc").requires_grad_(True); a = f("aw").requires_grad_(True)
out = MSDeformAttnFunction.apply(v, torch.from_numpy(d["shapes"]).to(DEV), torch.from_numpy(d["lsi"]).to(DEV), l, a, 64)
g = torch.autograd.grad(out, (v, l, a), f("grad_out"))
gl = g[1].detach().cpu().numpy().astype(np.float64)
diff = np.abs(gl - ref[2]); idx = np.argwhere(diff > 1e-3 * max(1, np.abs(ref[2]).max()))
print("n bad", len(idx))
for ix in idx[:6]:
    n, q, m, lv, p, c = ix
    x, y = np.float32(d["loc"][n, q, m, lv, p, 0]), np.float32(d["loc"][n, q, m, lv, p, 1])
    H, W = shapes[lv]
    print(tuple(ix), "loc", repr(x), repr(y), "H,W", H, W, "h_im32", repr(np.float32(y * np.float32(H)) - np.float32(0.5)), "w_im32", repr(np.float32(x * np.float32(W)) - np.float32(0.5)),
          "kernel", gl[tuple(ix)], "oracle32", ref[2][tuple(ix)], "oracle64", ref64[2][tuple(ix)])
