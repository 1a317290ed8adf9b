import os, sys
os.environ["MSDA_ENABLE_HOOKS"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import torch, numpy as np
import check_own as c
from devis_amd import _native
ms, a = c.run(1, {"MSDA_SCATTER_DBG": "512"}, reps=1)
ms, b = c.run(1, {"MSDA_SCATTER_DBG": "0"}, reps=1)
a = a.cpu().numpy(); b = b.cpu().numpy()
print("shape", a.shape, "nan in new:", np.isnan(b).sum(), "of", b.size)
bad = ~np.isclose(a, b, rtol=1e-4, atol=1e-6) 
print("bad elements", bad.sum())
idx = np.argwhere(bad)
print("frames", np.unique(idx[:,0]), "heads", np.unique(idx[:,2]))
pix = np.unique(idx[:,1]); print("n bad pixels", len(pix), pix[:40], pix[-10:])
ch = np.unique(idx[:,3]); print("channels", ch)
# per level
lsi = [0, 3600, 4520, 4760, 4820]
for l in range(4):
    sel = (idx[:,1] >= lsi[l]) & (idx[:,1] < lsi[l+1]); print("level", l, sel.sum())
i = idx[0]; print("example", i, a[tuple(i)], b[tuple(i)])
