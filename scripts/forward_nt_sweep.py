"""GPU probe: tiles per wave (workgroups per (clip, head)) of the resident-slab forward over batch sizes and storage types.

    python scripts/forward_nt_sweep.py
"""
import os
import sys

os.environ.setdefault("MSDA_ENABLE_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import scatter_ab


def knobs(**env):
    for k in ("MSDA_FWD_RS", "MSDA_FWD_RS_NT"):
        os.environ.pop(k, None)
    scatter_ab.knobs(**env)


def main():
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        for clips in (4, 8, 16, 32, 64):
            fwd, bwd, gv, reps = scatter_ab.temporal_case(clips, "A", "uniform", 300, dtype, 20)
            knobs()
            bench._event_ms(fwd, 5)                     # (warm: the first measurement of new tensors reads ~6 % slow)
            res = []
            for label, env in (("auto", {}), ("nt1", {"MSDA_FWD_RS": 1, "MSDA_FWD_RS_NT": 1}), ("nt2", {"MSDA_FWD_RS": 1, "MSDA_FWD_RS_NT": 2}),
                               ("nt4", {"MSDA_FWD_RS": 1, "MSDA_FWD_RS_NT": 4})):
                knobs(**env)
                res.append("%s %.4f" % (label, bench._event_ms(fwd, reps)))
            print("%2d clips %-8s forward: %s" % (clips, str(dtype)[6:], " | ".join(res)), flush=True)
            del fwd, bwd, gv
            torch.cuda.empty_cache()
    knobs()


if __name__ == "__main__":
    main()
