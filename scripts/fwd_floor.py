"""GPU probe (round 5): the headline forward and gather pass of whatever build MSDA_LIB names -- for the timing-only builds
-DMSDA_RS_EXP=1 (no memory corners), 2 (no LDS corners), 6 (neither: the skeleton = staging, barriers, point loads, geometry,
broadcasts, stores), 3 (no slab staging).  Results of those builds are wrong by construction."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from devis_amd import _native, tuning

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    out = []
    cases = [(torch.float32, "A", "decoder", 16), (torch.bfloat16, "A", "decoder", 16)]
    if os.environ.get("F16_CASES"):     # the f16 kernels (v_fma_mix_f32 rows, round 5): decoder batch, temporal and single-frame encoder calls
        cases = [(torch.float16, "A", "decoder", 16), (torch.float16, "S", "encoder", 1), (torch.float16, "B", "plain_encoder", 8),
                 (torch.float16, "S", "plain_decoder", 6)]
    if os.environ.get("ENC_CASES"):     # encoder-shaped calls (resident-window kernels on the 800x1333 pyramid)
        cases = [(torch.float32, "B", "encoder", 1), (torch.bfloat16, "B", "encoder", 1), (torch.float32, "A", "encoder", 1)]
    for dt, pyr, kind, clips in cases:
        fwd, bwd, d, so = tuning._case(tuning.PYRAMIDS[pyr], dt, dt, clips, 300, kind, 6, 8, 32, 4, dev)
        t_f = tuning._time(fwd, 21)
        r_f = _native.last_route()[14:58]
        os.environ["MSDA_ENABLE_HOOKS"] = "1"; os.environ["MSDA_BWD_PHASES"] = "1"
        _native.reload_knobs()
        t_g = tuning._time(bwd, 21)
        os.environ.pop("MSDA_BWD_PHASES"); _native.reload_knobs()
        out.append("%s %s %s fwd %.4f gather %.4f ms [%s]" % (str(dt).split(".")[1], pyr, kind, t_f, t_g, r_f[:28]))
        del fwd, bwd, so
        torch.cuda.empty_cache()
    print("%-40s %s" % (os.path.basename(os.environ.get("MSDA_LIB", "libmsda_hip.so")), "   ".join(out)), flush=True)
