"""GPU probe (round 5): the scatter of f16 calls (grad_out rows kept f16 in LDS and read with v_fma_mix_f32, MSDA_ROWS16=1) for the
build MSDA_LIB names; grad_value is checked against the fp32 run of the same (rounded) inputs."""
import os
import sys

os.environ["MSDA_ENABLE_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from devis_amd import _native, tuning

CASES = [("A", "decoder", 16, 300), ("A", "encoder", 1, 0), ("S", "encoder", 1, 0), ("B", "plain_encoder", 8, 0), ("S", "plain_decoder", 36, 300)]
if __name__ == "__main__":
    dev = torch.device("cuda:0")
    out = []
    for pyr, kind, clips, q in CASES:
        fwd, bwd, d, scatter_only = tuning._case(tuning.PYRAMIDS[pyr], torch.float16, torch.float16, clips, q, kind, 6, 8, 32, 4, dev)
        bwd()
        os.environ["MSDA_BWD_PHASES"] = "2"; _native.reload_knobs()
        t = tuning._time(scatter_only, 15)
        os.environ.pop("MSDA_BWD_PHASES"); _native.reload_knobs()
        out.append("%s %s x%d %.4f" % (pyr, kind, clips, t))
        del fwd, bwd, scatter_only
        torch.cuda.empty_cache()
    print("%-28s scatter ms: %s" % (os.path.basename(os.environ.get("MSDA_LIB", "libmsda_hip.so")), "   ".join(out)), flush=True)
