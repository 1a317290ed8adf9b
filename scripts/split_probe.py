"""GPU probe (round 5): the scatter of encoder-shaped calls with its heavy items split by query range (MSDA_SCATTER_SPLIT:
0 off, -1 the host's plan, n: n parts wherever the partial area allows) -- scatter-only times on the records of one backward."""
import os
import sys

os.environ["MSDA_ENABLE_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from devis_amd import _native, tuning

CASES = [("A", "encoder", 1, torch.float32), ("A", "encoder", 1, torch.bfloat16), ("B", "plain_encoder", 8, torch.bfloat16),
         ("B", "encoder", 1, torch.float32), ("S", "plain_encoder", 6, torch.float16), ("A", "encoder", 4, torch.float32)]
if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for pyr, kind, clips, dt in CASES:
        fwd, bwd, d, scatter_only = tuning._case(tuning.PYRAMIDS[pyr], dt, dt, clips, 0, kind, 6, 8, 32, 4, dev)
        line = []
        for split in ("0", "-1", "2", "3", "4", "8", "16"):
            os.environ["MSDA_SCATTER_SPLIT"] = split
            os.environ.pop("MSDA_BWD_PHASES", None)
            _native.reload_knobs()
            bwd()
            os.environ["MSDA_BWD_PHASES"] = "2"
            _native.reload_knobs()
            t = tuning._time(scatter_only, 9)
            line.append("%s: %.4f" % (split, t))
        os.environ.pop("MSDA_BWD_PHASES", None)
        print("%s %-13s clips %d %-8s scatter ms by MSDA_SCATTER_SPLIT  %s   [%s]" % (pyr, kind, clips, str(dt).split(".")[1], "  ".join(line),
              _native.last_route().split(";")[-2].strip()[-60:]), flush=True)
        del fwd, bwd, scatter_only
        torch.cuda.empty_cache()
