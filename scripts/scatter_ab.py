"""GPU probe: per-kernel times of the backward (gather pass, scatter) and of the forward on the shapes bench.py reports, for
ONE build of the library (MSDA_LIB selects it) -- the same-box A/B of scripts/ab_all.sh runs it once per build.  fp32 cases
also check grad_value against the LDS-atomic scatter of the same build (MSDA_SCATTER_OWN=0).

    python scripts/scatter_ab.py [case ...]        cases: dec16 dec16_bf16 dec1 encA encB cfg1 cfg4enc cfg4dec dec64
"""
import os
import sys

os.environ.setdefault("MSDA_ENABLE_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from devis_amd import _native

DEV = torch.device("cuda:0")
SWIN = [(60, 96), (30, 48), (15, 24), (8, 12)]
bench.PYRAMIDS["S"] = SWIN


def knobs(**env):
    for k in ("MSDA_BWD_PHASES", "MSDA_SCATTER_OWN", "MSDA_SCATTER_DBG"):
        os.environ.pop(k, None)
    # (probes run a gather pass under one route and scatter-only calls under others: records for every level, always)
    os.environ["MSDA_BWD_ALL_RECORDS"] = "1"
    os.environ.update({k: str(v) for k, v in env.items()})
    _native.reload_knobs()


def temporal_case(clips, pyr, locs, q, dtype, reps):
    class A:
        pass
    a = A()
    a.clips, a.frames, a.queries, a.pyramid, a.locs, a.sampling = clips, 6, q, pyr, locs, "storage"
    b = bench.make_clip_batch(a, DEV, dtype, 1)
    T, q, M, D, L, P, W, S = b["dims"]
    out = torch.empty((clips * T, q, M * D), dtype=dtype, device=DEV)
    gvt = _native.grad_value_dtype(b["value"], b["shapes"], q, L, P, clips=clips, window=W, Pt=P)
    gv = torch.zeros(b["value"].shape, dtype=gvt, device=DEV)
    grads = [torch.empty_like(b[k]) for k in ("loc_c", "aw_c", "loc_t", "aw_t")]
    ws = _native.bwd_workspace(DEV, clips * T, q, M, L * (1 + W))
    args = (b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"])

    def fwd():
        _native.temporal_forward(*args, clips, out)

    def bwd():
        ws[:16].zero_()         # the scatter's work tickets (zeroed by the gather pass; MSDA_BWD_PHASES=2 runs without one)
        _native.temporal_backward(*args, b["grad_out"], clips, gv, *grads, workspace=ws)
    return fwd, bwd, gv, reps


def plain_case(shapes, N, Lq, locs, dtype, reps):
    c = bench._plain_op_case(DEV, dtype, shapes, N, Lq, locs, seed=99)
    M, D, P, L = 8, 32, 4, c["L"]
    out = torch.empty((N, Lq, M * D), dtype=dtype, device=DEV)
    gv = torch.zeros(c["value"].shape, dtype=_native.grad_value_dtype(c["value"], c["shapes"], Lq, L, P), device=DEV)
    gl, ga = torch.empty_like(c["loc"]), torch.empty_like(c["aw"])
    ws = _native.bwd_workspace(DEV, N, Lq, M, L)
    lib = _native.load()

    def fwd():
        _native.forward(c["value"], c["shapes"], c["lsi"], c["loc"], c["aw"], out)

    def bwd():
        ws[:16].zero_()
        rc = lib.msda_backward(_native.dtype_code(dtype), c["value"].data_ptr(), c["shapes"].data_ptr(), c["lsi"].data_ptr(),
                               c["loc"].data_ptr(), c["aw"].data_ptr(), c["grad_out"].data_ptr(), N, c["S"], M, D, L, Lq, P,
                               gv.data_ptr(), _native.dtype_code(gv.dtype), gl.data_ptr(), ga.data_ptr(), ws.data_ptr(), ws.numel() * 4, None,
                               _native.shapes_hint(c["shapes"]), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.msda_last_error()
    return fwd, bwd, gv, reps


CASES = {
    "dec16": lambda: temporal_case(16, "A", "uniform", 300, torch.float32, 20),
    "dec16_bf16": lambda: temporal_case(16, "A", "uniform", 300, torch.bfloat16, 20),
    "dec64": lambda: temporal_case(64, "A", "uniform", 300, torch.float32, 6),
    "dec1": lambda: temporal_case(1, "A", "uniform", 300, torch.float32, 30),
    "dec2": lambda: temporal_case(2, "A", "uniform", 300, torch.float32, 30),
    "dec4": lambda: temporal_case(4, "A", "uniform", 300, torch.float32, 30),
    "dec8": lambda: temporal_case(8, "A", "uniform", 300, torch.float32, 20),
    "dec6S": lambda: temporal_case(6, "S", "uniform", 300, torch.float32, 20),
    "dec8B": lambda: temporal_case(8, "B", "uniform", 300, torch.float32, 10),
    "dec12_bf16": lambda: temporal_case(12, "A", "uniform", 300, torch.bfloat16, 20),
    "pdec36": lambda: plain_case(bench.PYRAMIDS["A"], 36, 300, "uniform", torch.float32, 20),
    "dec1_bf16": lambda: temporal_case(1, "A", "uniform", 300, torch.bfloat16, 30),
    "encA": lambda: temporal_case(1, "A", "local", 4820, torch.float32, 10),
    "encB": lambda: temporal_case(1, "B", "local", 22223, torch.float32, 5),
    "cfg1": lambda: plain_case(bench.PYRAMIDS["B"], 8, 22223, "local", torch.bfloat16, 8),
    "cfg4enc": lambda: plain_case(SWIN, 6, 7656, "local", torch.float16, 10),
    "cfg4dec": lambda: plain_case(SWIN, 6, 300, "uniform", torch.float16, 20),
}


def sweep(name):
    """MSDA_SCATTER_DBG ablations of the scatter on one case (timing only; results are wrong by construction)."""
    fwd, bwd, gv, reps = CASES[name]()
    knobs()
    bwd()
    out = []
    for label, dbg in (("full", 0), ("no walk", 1), ("no link", 2), ("no rows", 4), ("no walk/link/rows", 7), ("cull only", 8),
                       ("L0", 32), ("L1", 64), ("L2", 96), ("L3", 128), ("static order", 16)):
        knobs(MSDA_BWD_PHASES=2, MSDA_SCATTER_DBG=dbg)
        out.append("%s %.4f" % (label, bench._event_ms(bwd, reps)))
    knobs()
    print("%-22s %-11s sweep: %s" % (os.path.basename(os.environ.get("MSDA_LIB", "libmsda_hip.so")), name, " | ".join(out)), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "sweep":
        for name in sys.argv[2:]:
            sweep(name)
        return
    names = sys.argv[1:] or ["dec16", "dec16_bf16", "dec1", "encA", "encB", "cfg1", "cfg4enc"]
    tag = os.path.basename(os.environ.get("MSDA_LIB", "libmsda_hip.so"))
    for name in names:
        fwd, bwd, gv, reps = CASES[name]()
        knobs()
        t_f = bench._event_ms(fwd, reps)
        bwd()
        torch.cuda.synchronize()
        got = gv.float().clone()
        note = ""
        if gv.dtype == torch.float32 and name != "dec64":
            knobs(MSDA_SCATTER_OWN=0)
            bwd()
            torch.cuda.synchronize()
            ref = gv.clone()
            note = "  max|gv - lds-atomic| %.2e (scale %.2f)" % ((got - ref).abs().max().item(), ref.abs().max().item())
        knobs(MSDA_BWD_PHASES=1)
        t_g = bench._event_ms(bwd, reps)
        knobs(MSDA_BWD_PHASES=2)
        t_s = bench._event_ms(bwd, reps)
        if name in ("encA", "encB", "cfg1", "cfg4enc"):       # the other item order (MSDA_SCATTER_DBG=256: level by level, heaviest first)
            knobs(MSDA_BWD_PHASES=2, MSDA_SCATTER_DBG=256)
            note += "  scatter, level-by-level item order %.4f" % bench._event_ms(bwd, reps)
        knobs()
        print("%-22s %-11s fwd %.4f  gather %.4f  scatter %.4f ms%s" % (tag, name, t_f, t_g, t_s, note), flush=True)
        del fwd, bwd, gv, got
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
