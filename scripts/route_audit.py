"""GPU probe: is the AUTOMATIC route of the forward / gather pass the fastest one the library has, shape by shape?

    python scripts/route_audit.py [quick | enc | plain | scatter]

Temporal decoder calls (300 queries per frame) and encoder calls (every pixel a query, local sampling) on three pyramids -- 360x640,
SwinL 480x768 and 800x1333 -- at several batch sizes and storage types; every forced alternative that applies is timed after the
automatic choice (all after a warm-up).  A line ends with `<<` when a forced route beats the automatic one by more than 4 %.
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

bench.PYRAMIDS["S"] = scatter_ab.SWIN
KEYS = ("MSDA_SCATTER_DBG", "MSDA_SCATTER_OWN", "MSDA_FWD_RS", "MSDA_FWD_RS_NT", "MSDA_BWD_RS", "MSDA_BWD_RS_TPW", "MSDA_BWD_RS_FSPLIT", "MSDA_FWD_WIN", "MSDA_BWD_WIN")
FWD = (("tile", {"MSDA_FWD_RS": 0, "MSDA_FWD_WIN": 0}), ("rs1", {"MSDA_FWD_RS": 1, "MSDA_FWD_RS_NT": 1, "MSDA_FWD_WIN": 0}),
       ("rs2", {"MSDA_FWD_RS": 1, "MSDA_FWD_RS_NT": 2, "MSDA_FWD_WIN": 0}), ("rs4", {"MSDA_FWD_RS": 1, "MSDA_FWD_RS_NT": 4, "MSDA_FWD_WIN": 0}),
       ("win", {"MSDA_FWD_WIN": 1}))
BWD = (("tile", {"MSDA_BWD_RS": 0, "MSDA_BWD_WIN": 0}), ("rs1", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_TPW": 1, "MSDA_BWD_RS_FSPLIT": 0, "MSDA_BWD_WIN": 0}),
       ("rs2", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_TPW": 2, "MSDA_BWD_RS_FSPLIT": 0, "MSDA_BWD_WIN": 0}),
       ("rs4", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_TPW": 4, "MSDA_BWD_RS_FSPLIT": 0, "MSDA_BWD_WIN": 0}),
       ("fs2", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_FSPLIT": 2, "MSDA_BWD_WIN": 0}), ("fs4", {"MSDA_BWD_RS": 1, "MSDA_BWD_RS_FSPLIT": 4, "MSDA_BWD_WIN": 0}),
       ("win", {"MSDA_BWD_WIN": 1}))


def knobs(**env):
    for k in KEYS:
        os.environ.pop(k, None)
    scatter_ab.knobs(**env)


def audit(label, fn, alts, reps, extra):
    knobs(**extra)
    bench._event_ms(fn, 25)
    auto = bench._event_ms(fn, reps)
    route = _native.last_route()
    out, flag = [], ""
    for name, env in alts:
        knobs(**dict(env, **extra))
        try:
            fn()
            torch.cuda.synchronize()
            r = _native.last_route()
            if name == "win" and "window" not in r:
                continue
            t = bench._event_ms(fn, reps)
        except RuntimeError as exc:
            out.append("%s: %s" % (name, str(exc)[:30]))
            continue
        out.append("%s %.4f" % (name, t))
        if t < 0.96 * auto:
            flag = "  <<"
    knobs()
    print("%-34s auto %.4f [%s] | %s%s" % (label, auto, route.replace("msda ", "")[:60], " | ".join(out), flag), flush=True)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    cases = []
    for pyr in ("A", "S", "B"):
        for dtype in (torch.float32, torch.bfloat16):
            for clips in ((1, 16) if quick else (1, 2, 4, 8, 16, 32)):
                cases.append(("dec", pyr, clips, dtype))
            for clips in ((1,) if quick else ((1, 2) if pyr == "B" else (1, 2, 4, 8))):
                cases.append(("enc", pyr, clips, dtype))
    if len(sys.argv) > 1 and sys.argv[1] == "enc":
        cases = [c for c in cases if c[0] == "enc"]
    if len(sys.argv) > 1 and sys.argv[1] == "scatter":
        # grad_value scatter: automatic against the level-by-level item order and the static schedule
        alts = (("level order", {"MSDA_SCATTER_DBG": 256}), ("static", {"MSDA_SCATTER_DBG": 16}))
        todo = [("dec", pyr, c, dt) for pyr in ("A", "S", "B") for dt in (torch.float32, torch.bfloat16) for c in ((6, 8, 10, 12) if len(sys.argv) > 2 else (1, 4, 16, 32))]
        todo += [("enc", pyr, 1, dt) for pyr in ("A", "S", "B") for dt in (torch.float32, torch.bfloat16)]
        todo += [(kind, pyr, n, dt) for pyr in ("A", "S", "B") for dt in (torch.float32, torch.bfloat16)
                 for kind, ns in (("penc", (1, 8)), ("pdec", (6, 36))) for n in ns]
        for kind, pyr, clips, dtype in todo:
            S = sum(h * w for h, w in bench.PYRAMIDS[pyr])
            if kind in ("penc", "pdec"):
                fwd, bwd, gv, reps = scatter_ab.plain_case(bench.PYRAMIDS[pyr], clips, S if kind == "penc" else 300,
                                                           "local" if kind == "penc" else "uniform", dtype, 10)
            else:
                fwd, bwd, gv, reps = scatter_ab.temporal_case(clips, pyr, "local" if kind == "enc" else "uniform", S if kind == "enc" else 300, dtype, 10)
            knobs()
            bwd()
            audit("%s %s %2d %-8s scatter" % (kind, pyr, clips, str(dtype)[6:]), bwd,
                  alts, reps, {"MSDA_BWD_PHASES": 2})
            del fwd, bwd, gv
            torch.cuda.empty_cache()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "plain":
        # single-frame MSDeformAttn: the encoder call (every pixel a query, N images) and the decoder-like call (300 queries)
        cases = [(kind, pyr, n, dtype) for pyr in ("A", "S", "B") for dtype in (torch.float32, torch.bfloat16)
                 for kind, ns in (("penc", (1, 2, 6, 8)), ("pdec", (1, 6, 36))) for n in ns]
    for kind, pyr, clips, dtype in cases:
        S = sum(h * w for h, w in bench.PYRAMIDS[pyr])
        if kind in ("penc", "pdec"):
            fwd, bwd, gv, reps = scatter_ab.plain_case(bench.PYRAMIDS[pyr], clips, S if kind == "penc" else 300,
                                                       "local" if kind == "penc" else "uniform", dtype, 12)
            tag = "%s %s %2d images %-8s" % (kind, pyr, clips, str(dtype)[6:])
            audit(tag + " fwd", fwd, [a for a in FWD], reps, {})
            knobs()
            bwd()
            audit(tag + " gather", bwd, [a for a in BWD if not a[0].startswith("fs")], reps, {"MSDA_BWD_PHASES": 1})
            del fwd, bwd, gv
            torch.cuda.empty_cache()
            continue
        if kind == "enc" and clips * S * 6 * 256 * 4 * 12 > 40e9:
            continue
        try:
            fwd, bwd, gv, reps = scatter_ab.temporal_case(clips, pyr, "local" if kind == "enc" else "uniform", S if kind == "enc" else 300, dtype, 12)
        except Exception as exc:
            print(kind, pyr, clips, dtype, "skipped:", str(exc)[:80])
            continue
        tag = "%s %s %2d clips %-8s" % (kind, pyr, clips, str(dtype)[6:])
        audit(tag + " fwd", fwd, FWD, reps, {})
        knobs()
        bwd()
        audit(tag + " gather", bwd, BWD, reps, {"MSDA_BWD_PHASES": 1})
        del fwd, bwd, gv
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
