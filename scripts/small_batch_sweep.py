"""GPU probe: routes of the forward and of the gather pass on SMALL batches of the decoder call (1, 2, 4, 8 clips), per kernel.

    python scripts/small_batch_sweep.py [clips ...]

For every batch size: the automatic route, the tile kernels, the resident-slab kernels forced (1 / 2 / 4 tiles per wave) and the
gather pass with one source frame per workgroup (1, 2, 4, 8 workgroups per (clip, head, frame)).
"""
import os
import sys

os.environ.setdefault("MSDA_ENABLE_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import scatter_ab
from devis_amd import _native

KEYS = ("MSDA_FWD_RS", "MSDA_FWD_RS_NT", "MSDA_BWD_RS", "MSDA_BWD_RS_TPW", "MSDA_BWD_RS_FSPLIT")


def knobs(**env):
    for k in KEYS:
        os.environ.pop(k, None)
    scatter_ab.knobs(**env)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    for clips in sizes:
        for dtype in (torch.float32, torch.bfloat16):
            fwd, bwd, gv, reps = scatter_ab.temporal_case(clips, "A", "uniform", 300, dtype, 30)
            knobs()
            bwd()
            res = []
            for label, env in (("auto", {}), ("tile", {"MSDA_FWD_RS": 0}), ("rs nt1", {"MSDA_FWD_RS": 1, "MSDA_FWD_RS_NT": 1}),
                               ("rs nt2", {"MSDA_FWD_RS": 1, "MSDA_FWD_RS_NT": 2}), ("rs nt4", {"MSDA_FWD_RS": 1, "MSDA_FWD_RS_NT": 4})):
                knobs(**env)
                res.append("%s %.4f" % (label, bench._event_ms(fwd, reps)))
            print("%2d clips %-8s fwd:    %s" % (clips, str(dtype)[6:], " | ".join(res)), flush=True)
            res = []
            for label, env in (("auto", {}), ("tile", {"MSDA_BWD_RS": 0}),
                               ("rs tpw1", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_TPW": 1, "MSDA_BWD_RS_FSPLIT": 0}),
                               ("rs tpw2", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_TPW": 2, "MSDA_BWD_RS_FSPLIT": 0}),
                               ("fsplit1", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_FSPLIT": 1}), ("fsplit2", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_FSPLIT": 2}),
                               ("fsplit4", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_FSPLIT": 4}), ("fsplit8", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_FSPLIT": 8})):
                knobs(MSDA_BWD_PHASES=1, **env)
                res.append("%s %.4f" % (label, bench._event_ms(bwd, reps)))
            print("%2d clips %-8s gather: %s" % (clips, str(dtype)[6:], " | ".join(res)), flush=True)
            knobs(MSDA_BWD_PHASES=2)
            print("%2d clips %-8s scatter: %.4f" % (clips, str(dtype)[6:], bench._event_ms(bwd, reps)), flush=True)
            knobs()


if __name__ == "__main__":
    main()
