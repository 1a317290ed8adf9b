import os, sys
os.environ.setdefault("MSDA_ENABLE_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, scatter_ab
from devis_amd import _native
def knobs(**env):
    for k in ("MSDA_BWD_RS_FSPLIT", "MSDA_BWD_RS_TPW"):
        os.environ.pop(k, None)
    scatter_ab.knobs(**env)
for dtype in (torch.bfloat16, torch.float16, torch.float32):
    for clips in (8, 16, 32, 64):
        fwd, bwd, gv, reps = scatter_ab.temporal_case(clips, "A", "uniform", 300, dtype, 20)
        knobs(); bwd()
        res = []
        for label, env in (("auto", {}), ("fsplit0", {"MSDA_BWD_RS_FSPLIT": 0}), ("fsplit1", {"MSDA_BWD_RS_FSPLIT": 1}), ("fsplit2", {"MSDA_BWD_RS_FSPLIT": 2}), ("fsplit3", {"MSDA_BWD_RS_FSPLIT": 3})):
            knobs(MSDA_BWD_PHASES=1, **env)
            res.append("%s %.4f" % (label, bench._event_ms(bwd, reps)))
        print("%2d clips %-8s gather: %s" % (clips, str(dtype)[6:], " | ".join(res)), flush=True)
        del fwd, bwd, gv
        torch.cuda.empty_cache()
