"""GPU probe (round 6): where a wave of the matrix-pipe scatter kernel spends its clocks.  Needs a library built with
-DMSDA_MFMA_TRACE -DMSDA_TIMING_ONLY_BUILD (MSDA_LIB names it): every wave sums s_memtime deltas per phase of its items and leaves
them in the first eight level-0 pixels of its frame's grad_value (the kernel runs alone: MSDA_SCATTER_PART=2).  The stamps wait for
outstanding LDS / scalar-memory operations (s_memtime returns through lgkmcnt) and fence the scheduler: an attribution, not a timing."""
import os
import sys

os.environ["MSDA_ENABLE_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import scatter_ab as ab

NAMES = ("prologue", "load wait", "convert+advance+issue", "geometry+merge+cell writes", "products", "zero writes", "wait for the slowest wave",
         "reduction: barrier in front", "reduction: accumulators -> LDS", "reduction: barrier behind", "reduction: sums + stores", "last barrier")
NK = len(NAMES)
for name, dtype, clips in (("f32", torch.float32, 16), ("f32 x32", torch.float32, 32)):
    fwd, bwd, gv, reps = ab.temporal_case(clips, "A", "uniform", 300, dtype, 30)
    ab.knobs(MSDA_SCATTER_MFMA=1)
    bwd()
    ab.knobs(MSDA_SCATTER_MFMA=1, MSDA_BWD_PHASES=2, MSDA_SCATTER_PART=2)
    ms = bench._event_ms(bwd, reps, 5)
    gv.zero_()
    bwd()
    torch.cuda.synchronize()
    t = gv.view(torch.int32)[:, :8].reshape(gv.shape[0], 8, 8, 32).permute(0, 2, 1, 3).reshape(-1, 8, 32)[:, :, :16].double()   # [item, wave, k]
    os.environ.pop("MSDA_SCATTER_PART", None); os.environ.pop("MSDA_SCATTER_MFMA", None)
    ab.knobs()
    tot = t[:, :, :NK].sum(-1)
    print("%s: kernel alone %.4f ms; items %d; clocks per (item, wave): mean %.0f  max %.0f; steps per wave %.1f" %
          (name, ms, t.shape[0], tot.mean(), tot.max(), t[:, :, 15].mean()), flush=True)
    for k in range(NK):
        per_step = t[:, :, k].sum() / t[:, :, 15].sum()
        print("   %-28s %8.0f clk per (item, wave) = %5.1f %%   (%.0f per step)" % (NAMES[k], t[:, :, k].mean(), 100 * t[:, :, k].sum() / tot.sum(), per_step))
