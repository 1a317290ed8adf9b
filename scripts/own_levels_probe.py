"""GPU probe (round 6): the owner-computes scatter of the headline batch when it only handles the first n pyramid levels
(MSDA_SCATTER_OWN_LEVELS = n: measurement knob, the other levels' grad_value is not computed) -- what is left for it once the
matrix-pipe scatter takes the coarse levels.  Also dec16 in bf16 / f16."""
import os
import sys

os.environ["MSDA_ENABLE_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import scatter_ab as ab
from devis_amd import _native

for dtype in (torch.float32, torch.bfloat16, torch.float16):
    fwd, bwd, gv, reps = ab.temporal_case(16, "A", "uniform", 300, dtype, 30)
    out = []
    for n in (4, 3, 2, 1):
        ab.knobs(MSDA_BWD_PHASES=2, MSDA_SCATTER_OWN_LEVELS=n)
        out.append("levels [0,%d) %.4f" % (n, bench._event_ms(bwd, reps, 5)))
    ab.knobs()
    os.environ.pop("MSDA_SCATTER_OWN_LEVELS", None)
    _native.reload_knobs()
    print("%-9s scatter ms: %s" % (str(dtype).split(".")[1], "   ".join(out)), flush=True)
