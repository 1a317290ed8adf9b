"""GPU probe (round 6): the matrix-pipe scatter kernel ALONE (MSDA_SCATTER_PART=2) on the headline batch, for whatever build MSDA_LIB
names -- used with the timing-only builds -DMSDA_MFMA_EXP=1 (one product per step instead of 30), =4 (no loads inside the loop),
=5 (both) beside -DMSDA_TIMING_ONLY_BUILD, to see what the kernel's time is made of (profiles/NEGATIVE_RESULTS.md R6-6)."""
import os
import sys

os.environ["MSDA_ENABLE_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import scatter_ab as ab

out = []
for name, dtype, clips in (("f32", torch.float32, 16), ("bf16", torch.bfloat16, 16), ("f32 x32", torch.float32, 32)):
    fwd, bwd, gv, reps = ab.temporal_case(clips, "A", "uniform", 300, dtype, 30)
    ab.knobs(MSDA_SCATTER_MFMA=1)
    bwd()
    ab.knobs(MSDA_SCATTER_MFMA=1, MSDA_BWD_PHASES=2, MSDA_SCATTER_PART=2)
    out.append("%s %.4f" % (name, bench._event_ms(bwd, reps, 5)))
    os.environ.pop("MSDA_SCATTER_PART", None); os.environ.pop("MSDA_SCATTER_MFMA", None)
    ab.knobs()
print("%-26s matrix-pipe kernel alone, ms: %s" % (os.path.basename(os.environ.get("MSDA_LIB", "libmsda_hip.so")), "   ".join(out)), flush=True)
