"""GPU probe (round 5): resident-slab forward / gather pass of the 16-bit headline batch at every tiles-per-wave / frame-split route,
for whatever build MSDA_LIB names -- used to compare the shipped 1024-thread workgroups (one per CU) with two 512-thread workgroups
per CU on half the LDS each (-DMSDA_RS_THREADS=512 -DMSDA_RS_MAXFRAMES=8 -DMSDA_RS_LDS_BYTES=81920 -DMSDA_RS_MIN_WAVES=4)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from devis_amd import _native, tuning

FWD = (("rs1", {"MSDA_FWD_RS": 1, "MSDA_FWD_RS_NT": 1}), ("rs2", {"MSDA_FWD_RS": 1, "MSDA_FWD_RS_NT": 2}), ("rs4", {"MSDA_FWD_RS": 1, "MSDA_FWD_RS_NT": 4}))
GAT = (("rs1", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_TPW": 1, "MSDA_BWD_RS_FSPLIT": 0}), ("rs2", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_TPW": 2, "MSDA_BWD_RS_FSPLIT": 0}),
       ("fs2", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_FSPLIT": 2}), ("fs4", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_FSPLIT": 4}), ("fs8", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_FSPLIT": 8}))


def with_env(env, fn, reps=21):
    os.environ["MSDA_ENABLE_HOOKS"] = "1"
    for k, v in env.items():
        os.environ[k] = str(v)
    _native.reload_knobs()
    try:
        return tuning._time(fn, reps)
    finally:
        for k in env:
            os.environ.pop(k, None)
        _native.reload_knobs()


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    _native.clear_routes()
    tag = os.path.basename(os.environ.get("MSDA_LIB", "libmsda_hip.so"))
    # (round 6: with an argument, the fp32 headline batch -- two workgroups per CU then hold levels 2-3 only, level 1 comes from memory)
    cases = [(torch.float32, "A", "decoder", 16)] if len(sys.argv) > 1 else [(torch.bfloat16, "A", "decoder", 16), (torch.float16, "A", "decoder", 16), (torch.bfloat16, "A", "encoder", 1),
             (torch.float16, "S", "decoder", 6)]
    for dt, pyr, kind, clips in cases:
        fwd, bwd, d, so = tuning._case(tuning.PYRAMIDS[pyr], dt, dt, clips, 300, kind, 6, 8, 32, 4, dev)
        bwd()
        line = ["auto %.4f" % with_env({}, fwd)]
        for name, env in FWD:
            line.append("%s %.4f" % (name, with_env(env, fwd)))
        line.append("| gather auto %.4f" % with_env({"MSDA_BWD_PHASES": 1}, bwd))
        for name, env in GAT:
            line.append("%s %.4f" % (name, with_env(dict(env, MSDA_BWD_PHASES=1), bwd)))
        print("%-22s %s %s %s x%d: fwd %s" % (tag, str(dt).split(".")[1], pyr, kind, clips, " ".join(line)), flush=True)
        del fwd, bwd, so
        torch.cuda.empty_cache()
