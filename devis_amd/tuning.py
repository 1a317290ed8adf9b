"""Measured kernel routes: time the alternatives the library has for a call shape on THIS GPU and pin the winner
(include/msda.h, ``msda_pin_route``; round 5).

The library's route rules (``csrc/msda_api.hip``, ``launch_fast``) were calibrated on three pyramids and a few batch sizes.  They
stay as the fallback; ``tune`` replaces guessing by measurement for the shapes a user actually runs::

    import devis_amd
    devis_amd.tune([(45, 80), (23, 40), (12, 20), (6, 10)], torch.float32, clips=1, Lq=300, kind="decoder")

times the forward, the backward's gather pass and its scatter under every route that applies (resident-slab kernels at 1 / 2 / 4
tiles per wave, the frame-split gather grid, the resident-window kernels, the tile kernels, the scatter's item orders) on
synthetic inputs of that shape and pins, per direction, the fastest one if it beats the rules' choice by more than 5 %.  It
synchronises and allocates: call it once at start-up, outside any HIP-graph capture.  ``python -m devis_amd.tuning --audit`` runs
the shapes of DESIGN.md section 3.5 (three pyramids x batch sizes x storage types x call kinds) and writes the table that ships
as ``devis_amd/routes.json`` -- loaded with the library (``MSDA_ROUTES=0`` switches it off for A/B runs against the rules).

Kinds of call (``kind``):
  decoder        fused temporal call, ``Lq`` queries per frame, uniform sampling locations (TemporalMSDeformAttnDecoder)
  encoder        fused temporal call, every pixel a query (``Lq`` ignored), sampling round the query's own pixel (...Encoder)
  plain_decoder  single-frame ``MSDeformAttnFunction`` call, ``clips`` images, ``Lq`` queries, uniform locations
  plain_encoder  single-frame call, every pixel a query, local sampling (Deformable-DETR encoder; BASELINE configs[1])
"""
import json
import os
import sys

import torch

from . import _native

PYRAMIDS = {
    "A": [(45, 80), (23, 40), (12, 20), (6, 10)],        # 360x640 (DeVIS test size)
    "S": [(60, 96), (30, 48), (15, 24), (8, 12)],        # SwinL 480x768
    "B": [(100, 167), (50, 84), (25, 42), (13, 21)],     # 800x1333
}
_KNOB_ENV = {"fwd_rs": "MSDA_FWD_RS", "fwd_rs_nt": "MSDA_FWD_RS_NT", "fwd_win": "MSDA_FWD_WIN", "fwd_tile_waves": "MSDA_FWD_TILE_WAVES",
             "bwd_rs": "MSDA_BWD_RS", "bwd_rs_tpw": "MSDA_BWD_RS_TPW", "bwd_rs_fsplit": "MSDA_BWD_RS_FSPLIT", "bwd_win": "MSDA_BWD_WIN",
             "scatter_mfma": "MSDA_SCATTER_MFMA"}
FWD_ROUTES = (("tile", {"fwd_rs": 0, "fwd_win": 0}), ("tile3", {"fwd_rs": 0, "fwd_win": 0, "fwd_tile_waves": 3}),
              ("rs1", {"fwd_rs": 1, "fwd_rs_nt": 1, "fwd_win": 0}), ("rs2", {"fwd_rs": 1, "fwd_rs_nt": 2, "fwd_win": 0}),
              ("rs4", {"fwd_rs": 1, "fwd_rs_nt": 4, "fwd_win": 0}), ("win", {"fwd_win": 1}))
GATHER_ROUTES = (("tile", {"bwd_rs": 0, "bwd_win": 0}), ("rs1", {"bwd_rs": 1, "bwd_rs_tpw": 1, "bwd_rs_fsplit": 0, "bwd_win": 0}),
                 ("rs2", {"bwd_rs": 1, "bwd_rs_tpw": 2, "bwd_rs_fsplit": 0, "bwd_win": 0}),
                 ("rs4", {"bwd_rs": 1, "bwd_rs_tpw": 4, "bwd_rs_fsplit": 0, "bwd_win": 0}),
                 ("fs2", {"bwd_rs": 1, "bwd_rs_fsplit": 2, "bwd_win": 0}), ("fs4", {"bwd_rs": 1, "bwd_rs_fsplit": 4, "bwd_win": 0}),
                 ("win", {"bwd_win": 1}))
# (round 6: + the matrix-pipe scatter of the coarse levels forced on / off, alone and with the image order)
SCATTER_ROUTES = (("levels", {"scatter_order": 1}), ("image", {"scatter_order": 2}), ("mfma", {"scatter_mfma": 1}),
                  ("nomfma", {"scatter_mfma": 0}), ("image+mfma", {"scatter_order": 2, "scatter_mfma": 1}))
MARGIN = 0.95           # an alternative is pinned only when it takes less than this fraction of the rules' time (3 % pins flipped between two audits)


class _Knobs:
    """Route knobs through the environment (MSDA_ENABLE_HOOKS=1 + msda_reload_knobs), restored on exit."""

    def __init__(self):
        self._names = ["MSDA_ENABLE_HOOKS", "MSDA_BWD_PHASES", "MSDA_SCATTER_DBG"] + list(_KNOB_ENV.values())
        self._saved = None

    def __enter__(self):
        self._saved = {k: os.environ.get(k) for k in self._names}
        return self

    def set(self, settings=None, phases=None):
        for k in self._names:
            os.environ.pop(k, None)
        os.environ["MSDA_ENABLE_HOOKS"] = "1"
        for name, v in (settings or {}).items():
            if name == "scatter_order":
                os.environ["MSDA_SCATTER_DBG"] = str({1: 256, 2: 2048}[int(v)])
            else:
                os.environ[_KNOB_ENV[name]] = str(int(v))
        if phases is not None:
            os.environ["MSDA_BWD_PHASES"] = str(phases)
        _native.reload_knobs()

    def __exit__(self, *exc):
        for k, v in self._saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _native.reload_knobs()
        return False


def _centres(shapes):
    return torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij"), -1)
                      .reshape(-1, 2).flip(-1) for h, w in shapes], 0)            # [S, 2] as (x, y)


def _case(shapes, dtype, loc_dtype, clips, Lq, kind, frames, heads, channels, points, device, seed=2024):
    """Synthetic inputs of one call + closures that run its forward and its backward through the C ABI."""
    g = torch.Generator(device=device).manual_seed(seed)        # (on the device: the 800x1333 encoder call has 170 M location values)
    rand = lambda *s: torch.rand(*s, generator=g, device=device)
    randn = lambda *s: torch.randn(*s, generator=g, device=device)
    sh = torch.tensor(shapes, dtype=torch.int64)
    L, S = sh.shape[0], int(sh.prod(1).sum())
    temporal = kind in ("decoder", "encoder")
    T = frames if temporal else 1
    W = T - 1 if temporal else 0
    local = kind in ("encoder", "plain_encoder")
    q = S if local else Lq
    G, M, D, P = clips * T, heads, channels, points
    wh = torch.stack([sh[:, 1], sh[:, 0]], -1).float()
    mk = lambda x: x.to(device=device, dtype=dtype).contiguous()
    mkl = lambda x: x.to(device=device, dtype=loc_dtype).contiguous()       # sampling locations / attention weights

    def locations(levels, wh_levels):
        if not local:
            return rand(G, q, M, levels, P, 2)
        ref = _centres(shapes).to(device)[None, :, None, None, None, :]
        return ref + randn(G, q, M, levels, P, 2) * 2.0 / wh_levels.to(device)[None, None, None, :, None, :]

    value = mk(rand(G, S, M, D) * 2 - 1)
    loc_c = mkl(locations(L, wh))
    aw = torch.softmax(randn(G, q, M, L * P + W * L * P), -1)
    aw_c = mkl(aw[..., :L * P].reshape(G, q, M, L, P))
    go = mk(randn(G, q, M * D))
    shapes_d = sh.to(device)
    lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1])).to(device)
    out = torch.empty(G, q, M * D, dtype=dtype, device=device)
    gv = torch.empty(value.shape, dtype=torch.float32, device=device)
    glc, gac = torch.empty_like(loc_c), torch.empty_like(aw_c)
    ws = _native.bwd_workspace(device, G, q, M, L * (1 + W))
    if temporal:
        loc_t = mkl(locations(W * L, wh.repeat(W, 1)))
        aw_t = mkl(aw[..., L * P:].reshape(G, q, M, W * L, P))
        ftab = torch.tensor([[f for f in range(T) if f != t] for t in range(T)], dtype=torch.int32, device=device)
        glt, gat = torch.empty_like(loc_t), torch.empty_like(aw_t)
        fwd = lambda: _native.temporal_forward(value, shapes_d, lsi, ftab, loc_c, aw_c, loc_t, aw_t, clips, out)
        bwd = lambda: _native.temporal_backward(value, shapes_d, lsi, ftab, loc_c, aw_c, loc_t, aw_t, go, clips, gv, glc, gac, glt, gat, ws)
    else:
        fwd = lambda: _native.forward(value, shapes_d, lsi, loc_c, aw_c, out)

        def bwd():
            N, _, _, _ = value.shape
            rc = _native.load().msda_backward(_native.type_code(value.dtype, loc_c.dtype), value.data_ptr(), shapes_d.data_ptr(),
                                              lsi.data_ptr(), loc_c.data_ptr(), aw_c.data_ptr(), go.data_ptr(), N, S, M, D, L, q, P,
                                              gv.data_ptr(), 0, glc.data_ptr(), gac.data_ptr(), ws.data_ptr(), ws.numel() * 4, None,
                                              _native.shapes_hint(shapes_d), torch.cuda.current_stream().cuda_stream)
            if rc:
                raise RuntimeError(_native.load().msda_last_error().decode())
    dims = dict(clips=clips, frames=T, window=W, S=S, M=M, D=D, L=L, Lq=q, Pc=P, Pt=P if temporal else 1)

    def scatter_only():
        # MSDA_BWD_PHASES=2 runs the scatter alone on the records the last gather pass left; the gather pass is also what zeroes
        # the scatter's work-ticket counters (include/msda.h), so they are zeroed here (a 64-byte fill, part of every timing alike)
        ws[:16].zero_()
        bwd()
    return fwd, bwd, dims, scatter_only


def _time(fn, reps):
    """Median of `reps` event-timed calls (ms), after two warm-up calls."""
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def _race(knobs, fn, alternatives, phases, reps, accept):
    """Time the rules' choice and every alternative that really takes another route; returns (auto ms, {name: ms}, best)."""
    knobs.set(None, phases)
    auto = _time(fn, reps)
    times, best = {}, (None, auto)
    for name, settings in alternatives:
        knobs.set(settings, phases)
        try:
            fn()
            torch.cuda.synchronize()
            route = _native.last_route()
            if not accept(name, route):
                continue
            t = _time(fn, reps)
        except RuntimeError:
            continue
        times[name] = round(t, 5)
        if t < best[1]:
            best = (name, t)
    # (a second look at a would-be winner: both once more, interleaved, so that a clock ramp does not pick routes)
    if best[0] is not None and best[1] < MARGIN * auto:
        settings = dict(alternatives)[best[0]]
        knobs.set(None, phases); a2 = _time(fn, reps)
        knobs.set(settings, phases); b2 = _time(fn, reps)
        auto, best = min(auto, a2), (best[0], min(best[1], b2))
    return auto, times, best


def _accept(name, route):
    if name == "win":
        return "window" in route
    if name.startswith("rs") or name.startswith("fs"):
        return "resident-slab" in route
    return True


def tune(spatial_shapes, dtype=torch.float32, clips=1, Lq=300, kind="decoder", frames=6, heads=8, channels=32, points=4,
         device=None, reps=15, pin=True, verbose=False, sampling_fp32=False, keep_pins=False):
    """Time this call shape's routes, pin the winners (``pin=True``; ``keep_pins=True, pin=False``: leave the table alone and
    report how the CURRENT choice -- rules + pins -- compares with every alternative); returns a report::

        {"forward": {"key":…, "auto_ms":…, "times": {route: ms}, "pinned": {...} | None}, "backward": {...}}
    """
    if kind not in ("decoder", "encoder", "plain_decoder", "plain_encoder"):
        raise ValueError("kind: decoder | encoder | plain_decoder | plain_encoder")
    if not torch.cuda.is_available():
        raise RuntimeError("devis_amd.tune: needs the GPU it tunes for")
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("devis_amd.tune: not inside a HIP-graph capture (it synchronises)")
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    shapes = [tuple(int(v) for v in hw) for hw in (spatial_shapes.tolist() if hasattr(spatial_shapes, "tolist") else spatial_shapes)]
    # (16-bit modules hand the operator float32 sampling locations by default -- `sampling_fp32`, ABI v11 -- a call shape of its own)
    loc_dtype = torch.float32 if (sampling_fp32 and dtype in (torch.bfloat16, torch.float16)) else dtype
    fwd, bwd, d, scatter_only = _case(shapes, dtype, loc_dtype, clips, Lq, kind, frames, heads, channels, points, device)
    code = _native.type_code(dtype, loc_dtype)
    keys = {b: _native.route_key(b, code, d["clips"], d["frames"], d["window"], d["S"], d["M"], d["D"], d["L"], d["Lq"], d["Pc"], d["Pt"],
                                 shapes) for b in (False, True)}
    if keep_pins and pin:
        raise ValueError("keep_pins measures the current pins against the alternatives; it does not re-pin")
    if not keep_pins:
        for k in keys.values():
            _native.pin_route(k, "")                # measure against the RULES, not against an older pin
    report = {}
    with _Knobs() as knobs:
        auto, times, best = _race(knobs, fwd, FWD_ROUTES, None, reps, _accept)
        chosen = dict(dict(FWD_ROUTES)[best[0]]) if best[0] is not None and best[1] < MARGIN * auto else None
        report["forward"] = {"key": keys[False], "auto_ms": round(auto, 5), "times": times, "best": best[0], "pinned": chosen}
        g_auto, g_times, g_best = _race(knobs, bwd, GATHER_ROUTES, 1, reps, _accept)
        knobs.set(None, None)
        bwd()                                       # (a full backward first: the scatter-only calls below read its culling records)
        s_auto, s_times, s_best = _race(knobs, scatter_only, SCATTER_ROUTES, 2, reps, lambda n, r: "owner-computes" in r)
        chosen_b = {}
        if g_best[0] is not None and g_best[1] < MARGIN * g_auto:
            chosen_b.update(dict(GATHER_ROUTES)[g_best[0]])
        if s_best[0] is not None and s_best[1] < MARGIN * s_auto:
            chosen_b.update(dict(SCATTER_ROUTES)[s_best[0]])
        report["backward"] = {"key": keys[True], "gather_auto_ms": round(g_auto, 5), "gather_times": g_times, "gather_best": g_best[0],
                              "scatter_auto_ms": round(s_auto, 5), "scatter_times": s_times, "scatter_best": s_best[0],
                              "pinned": chosen_b or None}
    if pin:
        if report["forward"]["pinned"]:
            _native.pin_route(keys[False], report["forward"]["pinned"])
        if report["backward"]["pinned"]:
            _native.pin_route(keys[True], report["backward"]["pinned"])
    if verbose:
        f, b = report["forward"], report["backward"]
        gather_pinned = bool(b["pinned"]) and any(k.startswith("bwd_") for k in b["pinned"])
        scatter_pinned = bool(b["pinned"]) and ("scatter_order" in b["pinned"] or "scatter_mfma" in b["pinned"])
        print("%-14s %-5s%s clips %-3d Lq %-6d %s | fwd %.4f %s -> %s | gather %.4f %s -> %s | scatter %.4f %s -> %s" % (
            kind, str(dtype).split(".")[1], "+loc32" if loc_dtype != dtype else "", clips, d["Lq"], "x".join(str(v) for v in shapes[0]),
            f["auto_ms"], f["times"], f["best"] if f["pinned"] else "-",
            b["gather_auto_ms"], b["gather_times"], b["gather_best"] if gather_pinned else "-",
            b["scatter_auto_ms"], b["scatter_times"], b["scatter_best"] if scatter_pinned else "-"), flush=True)
    return report


def audit_shapes(quick=False):
    """The shapes of the shipped table: (pyramid, kind, clips, Lq, dtype, float32 sampling locations beside a 16-bit value)."""
    out = []
    for pyr in ("A", "S", "B"):
        for dt, l32 in ((torch.float32, False), (torch.bfloat16, False), (torch.float16, False), (torch.bfloat16, True), (torch.float16, True)):
            if quick and (dt == torch.float16 or l32):
                continue
            for clips in ((1, 16) if quick else (1, 2, 4, 8, 16, 32)):
                out.append((pyr, "decoder", clips, 300, dt, l32))
            if not quick:
                for q in (60, 180):
                    out.append((pyr, "decoder", 1, q, dt, l32))
            for clips in ((1,) if (quick or pyr == "B") else (1, 2, 4)):
                out.append((pyr, "encoder", clips, 0, dt, l32))
            for n in ((8,) if quick else (1, 6, 8)):
                out.append((pyr, "plain_encoder", n, 0, dt, l32))
            for n in ((36,) if quick else (6, 36)):
                out.append((pyr, "plain_decoder", n, 300, dt, l32))
    return out


def audit(path, quick=False, reps=15):
    """Tune every shape of :func:`audit_shapes`; write {"device":…, "routes": {key: settings}, "log": [...]} to ``path``."""
    routes, log = {}, []
    for pyr, kind, clips, q, dt, l32 in audit_shapes(quick):
        if l32 and kind.startswith("plain"):
            continue                                 # (the plain module computes its locations in the storage type)
        try:
            r = tune(PYRAMIDS[pyr], dt, clips=clips, Lq=q, kind=kind, reps=reps, pin=False, verbose=True, sampling_fp32=l32)
        except (RuntimeError, torch.cuda.OutOfMemoryError) as e:       # a shape this GPU cannot hold: skipped, said so
            print("skipped %s %s clips %d %s: %s" % (pyr, kind, clips, dt, str(e)[:80]), flush=True)
            continue
        finally:
            torch.cuda.empty_cache()
        for side in ("forward", "backward"):
            if r[side]["pinned"]:
                routes[r[side]["key"]] = r[side]["pinned"]
        log.append({"pyramid": pyr, "kind": kind, "clips": clips, "Lq": q, "dtype": str(dt).split(".")[1] + ("+loc32" if l32 else ""), "report": r})
    doc = {"device": torch.cuda.get_device_name(0), "margin": MARGIN, "routes": routes, "log": log}
    with open(path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print("%d shapes, %d pinned routes -> %s" % (len(log), len(routes), path))
    return doc


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--audit", action="store_true", help="tune the shipped table's shapes")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "routes.json"))
    ap.add_argument("--reps", type=int, default=15)
    a = ap.parse_args()
    if not a.audit:
        sys.exit("usage: python -m devis_amd.tuning --audit [--quick] [--out routes.json]")
    audit(a.out, a.quick, a.reps)
