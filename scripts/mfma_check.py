"""GPU probe (round 6): the matrix-pipe scatter of the coarse levels (MSDA_SCATTER_MFMA=1) against the owner-computes scatter of the
same build (=0) -- grad_value differences and the scatter's time, on the shapes bench.py reports.

    python scripts/mfma_check.py [case ...]      cases: dec16 dec16_bf16 dec16_f16 dec4 dec1 decS decB plain
"""
import os
import sys

os.environ["MSDA_ENABLE_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import scatter_ab as ab
from devis_amd import _native

CASES = {
    "dec16": lambda: ab.temporal_case(16, "A", "uniform", 300, torch.float32, 30),
    "dec16_bf16": lambda: ab.temporal_case(16, "A", "uniform", 300, torch.bfloat16, 30),
    "dec16_f16": lambda: ab.temporal_case(16, "A", "uniform", 300, torch.float16, 30),
    "dec16_clu": lambda: ab.temporal_case(16, "A", "clustered", 300, torch.float32, 30),
    "dec4": lambda: ab.temporal_case(4, "A", "uniform", 300, torch.float32, 30),
    "dec1": lambda: ab.temporal_case(1, "A", "uniform", 300, torch.float32, 30),
    "dec64": lambda: ab.temporal_case(64, "A", "uniform", 300, torch.float32, 10),
    "decS": lambda: ab.temporal_case(16, "S", "uniform", 300, torch.float32, 20),
    "decB": lambda: ab.temporal_case(8, "B", "uniform", 300, torch.float32, 20),
    "dec60": lambda: ab.temporal_case(16, "A", "uniform", 60, torch.float32, 30),
    "plain": lambda: ab.plain_case(bench.PYRAMIDS["A"], 48, 300, "uniform", torch.float32, 30),
    "dec8": lambda: ab.temporal_case(8, "A", "uniform", 300, torch.float32, 30),
    "dec2": lambda: ab.temporal_case(2, "A", "uniform", 300, torch.float32, 30),
    "dec32": lambda: ab.temporal_case(32, "A", "uniform", 300, torch.float32, 10),
    "dec180": lambda: ab.temporal_case(16, "A", "uniform", 180, torch.float32, 30),
    # encoder-shaped calls (one query per pixel): few, very long items -- outside the automatic rule, forced here for the record
    "encA": ab.CASES["encA"], "encA4": lambda: ab.temporal_case(4, "A", "local", 4820, torch.float32, 6),
    "encB": ab.CASES["encB"], "cfg1": ab.CASES["cfg1"], "cfg4enc": ab.CASES["cfg4enc"],
}


def main():
    names = sys.argv[1:] or ["dec16", "dec16_bf16", "dec16_f16", "dec16_clu", "dec4", "dec1", "decS", "decB", "dec60", "plain"]
    for name in names:
        fwd, bwd, gv, reps = CASES[name]()
        res = {}
        for mode in (0, 1):
            ab.knobs(MSDA_SCATTER_MFMA=mode)
            gv.fill_(float("nan"))
            bwd()
            torch.cuda.synchronize()
            route = _native.last_route()
            res[mode] = gv.float().clone()
            ab.knobs(MSDA_SCATTER_MFMA=mode, MSDA_BWD_PHASES=2)
            t = bench._event_ms(bwd, reps, 5)
            res[("t", mode)] = t
            res[("r", mode)] = route.split("; ")[-1]
        # the owner-computes kernel alone on the levels the matrix-pipe kernel leaves it (measurement knob; the difference to the
        # mfma=1 time is the matrix-pipe kernel's own duration)
        lv = 2 if "two" in os.environ.get("MFMA_OWN", "two") else 3
        own = {}
        for n in (3, 2):
            ab.knobs(MSDA_SCATTER_MFMA=0, MSDA_BWD_PHASES=2, MSDA_SCATTER_OWN_LEVELS=n)
            own[n] = bench._event_ms(bwd, reps, 5)
        os.environ.pop("MSDA_SCATTER_MFMA", None); os.environ.pop("MSDA_SCATTER_OWN_LEVELS", None)
        ab.knobs()
        ref, got = res[0], res[1]
        err = float((got - ref).abs().max())
        scale = float(ref.abs().max())
        nan = int(torch.isnan(got).sum())
        print("%-11s scatter %.4f -> %.4f ms   max|diff| %.3e of scale %.3e = %.2e   nan %d   owner kernel on levels [0,3) %.4f [0,2) %.4f"
              % (name, res[("t", 0)], res[("t", 1)], err, scale, err / max(scale, 1e-30), nan, own[3], own[2]), flush=True)


if __name__ == "__main__":
    main()
