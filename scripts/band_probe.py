"""GPU probe (round 5): gather pass of the headline batch (fp32, bf16) at the frame-split grids named in FSPLITS, for the build MSDA_LIB
names -- the timing-only gate of the level-0-band idea (profiles/NEGATIVE_RESULTS.md R5-16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devis_amd import _native, tuning
dev = torch.device("cuda:0")
_native.clear_routes()
for dt in (torch.float32, torch.bfloat16):
    fwd, bwd, d, so = tuning._case(tuning.PYRAMIDS["A"], dt, dt, 16, 300, "decoder", 6, 8, 32, 4, dev)
    bwd()
    out = []
    for fs in os.environ.get("FSPLITS", "-1").split(","):
        os.environ["MSDA_ENABLE_HOOKS"] = "1"; os.environ["MSDA_BWD_PHASES"] = "1"; os.environ["MSDA_BWD_RS"] = "1"
        if fs != "-1": os.environ["MSDA_BWD_RS_FSPLIT"] = fs
        _native.reload_knobs()
        t = tuning._time(bwd, 21)
        out.append("fsplit %s: %.4f ms [%s]" % (fs, t, _native.last_route()[-60:]))
        for k in ("MSDA_BWD_PHASES", "MSDA_BWD_RS_FSPLIT", "MSDA_BWD_RS"): os.environ.pop(k, None)
        _native.reload_knobs()
    print("%-22s %s gather pass: %s" % (os.path.basename(os.environ.get("MSDA_LIB", "libmsda_hip.so")), str(dt).split(".")[1], " | ".join(out)), flush=True)
    del fwd, bwd, so
