"""GPU probe (round 5): resident-slab forward on pyramids whose slab starts at level 2 (SwinL in 4-byte types, 800x1333), one tile
per wave: the software-pipelined slot body compiled for a level-2 slab (MSDA_FWD_RS_BODY=2) against the plain point loop the
kernel compiled for a level-1 slab falls back to (=1).  The fp32 and f16 level-2 bodies spill 31 / 24 VGPRs (profiles/r05_resource_usage.txt)."""
import os
import sys

os.environ["MSDA_ENABLE_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from devis_amd import _native, tuning

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for pyr in ("B", "S"):
        for dt in (torch.float16, torch.bfloat16, torch.float32):
            for clips in (2, 4, 8, 16):
                fwd, bwd, d, so = tuning._case(tuning.PYRAMIDS[pyr], dt, dt, clips, 300, "decoder", 6, 8, 32, 4, dev)
                res = []
                for nt in ("1", "2"):
                    for body in ("1", "2"):
                        os.environ.update({"MSDA_FWD_RS": "1", "MSDA_FWD_RS_NT": nt, "MSDA_FWD_RS_BODY": body, "MSDA_FWD_WIN": "0"})
                        _native.reload_knobs()
                        res.append("nt%s body%s %.4f" % (nt, body, tuning._time(fwd, 11)))
                for k in ("MSDA_FWD_RS", "MSDA_FWD_RS_NT", "MSDA_FWD_RS_BODY", "MSDA_FWD_WIN"):
                    os.environ.pop(k)
                _native.reload_knobs()
                res.append("auto %.4f [%s]" % (tuning._time(fwd, 11), _native.last_route()[14:60]))
                print("%s %-8s clips %-2d  %s" % (pyr, str(dt).split(".")[1], clips, "  ".join(res)), flush=True)
                del fwd, bwd, so
                torch.cuda.empty_cache()
