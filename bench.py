#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MSDeformAttn hot path on MI355X.

Metric (BASELINE.json): MSDeformAttn fwd+bwd M-queries/s at T=6, L=4, K=4, C=256 (configs[2], the
DeVIS decoder temporal attention: T=6 frames, 300 queries per frame, pyramid of the 360x640 DeVIS
test size, M=8 heads x D=32).  One "step" = forward + backward of one decoder-layer temporal
attention over a batch of `--clips` independent clips (synthetic, seeded), inputs resident in HBM.
A query row = one (frame, query) producing C=256 outputs; M-queries/s = clips*T*300 / step time / 1e6.

    python bench.py                       # 1 GPU, default K/W, prints ONE JSON line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          # no launcher: bench.py starts that very torchrun as a child (launch_ranks),
                                          # relays its line and exits with its status; < N devices visible = an error

Multi-GPU (`--gpus N`): clip-parallel -- every rank runs the same per-GPU batch of clips (weak
scaling), no collective on the data path; timing = barrier + synchronize on both sides, max over ranks.

The JSON line also carries
  roofline     -- for the dominant kernel (the longer of the fused forward / backward kernels):
                  algorithmic bytes per launch / its average launch duration measured
                  here with HIP events on the launch stream, against the 8 TB/s HBM peak;
  cpu_baseline -- the reference's pure-PyTorch CPU path (oracle/ restatement of
                  ms_deform_attn_core_pytorch, fwd+bwd through autograd, reference call pattern)
                  timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PYRAMIDS = {
    "A": [(45, 80), (23, 40), (12, 20), (6, 10)],        # 360x640 (DeVIS test size), S = 4820
    "B": [(100, 167), (50, 84), (25, 42), (13, 21)],     # 800x1333, S = 22223
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable with a float4 copy)
DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--prewarm-seconds", type=float, default=0.5,
                    help="untimed device warm-up (clocks, allocator) before the W warm-up steps")
    ap.add_argument("--clips", type=int, default=16, help="clips per GPU per step")
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--queries", type=int, default=300, help="queries per frame")
    ap.add_argument("--pyramid", choices=sorted(PYRAMIDS), default="A")
    ap.add_argument("--dtype", choices=sorted(DTYPES), default="f32")
    ap.add_argument("--sampling", choices=["storage", "fp32"], default="storage",
                    help="dtype of sampling locations / attention weights beside a 16-bit value: storage = the same 16-bit "
                         "type (the reference's single-dtype contract); fp32 = float32 (ABI v11 MSDA_*_LOC32, what devis_amd's "
                         "modules feed the op by default)")
    ap.add_argument("--locs", choices=["uniform", "clustered", "local"], default="uniform",
                    help="uniform: rand in [0,1) as the reference test.py; clustered: reference point + "
                         "N(0, (3 px)^2) offsets per level, as a trained decoder produces; local: query i sits on "
                         "pixel i of the pyramid (needs --queries S) and samples N(0, (2 px)^2) around it in every "
                         "frame, as the temporal ENCODER does")
    ap.add_argument("--sigma", type=float, default=0.0, help="spread in pixels of the clustered / local sampling (0: 3 px clustered, 2 px local)")
    ap.add_argument("--mode", choices=["clip-parallel", "sharded"], default="clip-parallel",
                    help="clip-parallel: independent clips per GPU, no collective (default, weak scaling); "
                         "sharded: every clip is split over ALL ranks (devis_amd/clip_parallel.py: RCCL all-gather "
                         "of value forward, reduce-scatter of grad_value backward; strong scaling)")
    ap.add_argument("--transport", choices=["storage", "bf16", "f16"], default="storage",
                    help="--mode sharded: dtype in which `value` crosses the all-gather and its gradient the reduce-scatter "
                         "(storage = the model's own; bf16 / f16 with an f32 model: half the xGMI bytes, clip_parallel.py)")
    ap.add_argument("--pattern", choices=["fused", "reference"], default="fused",
                    help="fused: one launch per direction; reference: the 2*T calls per layer of the reference")
    ap.add_argument("--value-layout", choices=["dense", "padded"], default="dense",
                    help="dense = the reference's [N,S,M,D]; padded = one spare head slot per pixel row, what "
                         "devis_amd's modules feed the op (functions.project_value; DESIGN.md section 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the extra keys for the other BASELINE configs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of an N > 1 run (nccl = RCCL; gloo only for --share-device test runs)")
    ap.add_argument("--share-device", action="store_true",
                    help="TEST ONLY: every rank uses cuda:0 (exercises the N-rank protocol -- rendezvous, barriers, max-over-ranks, "
                         "ranks_seen -- on a box with one GPU; RCCL refuses two ranks on one device, so it needs --dist-backend gloo; "
                         "the line says so and its value is not a scaling figure)")
    return ap.parse_args()


def make_clip_batch(args, device, dtype, seed):
    """Synthetic decoder-layer inputs for `clips` clips.  Seeded; joint softmax over the 96 logits."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    T, q, M, D, P = args.frames, args.queries, 8, 32, 4
    shapes = torch.tensor(PYRAMIDS[args.pyramid], dtype=torch.int64)
    L, W = shapes.shape[0], T - 1
    S = int(shapes.prod(1).sum())
    G = args.clips * T
    value = (torch.rand(G, S, M, D, generator=g) * 2 - 1)
    if args.locs == "uniform":
        loc_c = torch.rand(G, q, M, L, P, 2, generator=g)
        loc_t = torch.rand(G, q, M, W * L, P, 2, generator=g)
    else:
        sigma = 3.0
        ref = torch.rand(G, q, 1, 1, 1, 2, generator=g)
        if args.locs == "local":
            if q != S:
                sys.exit("--locs local needs --queries %d (= S of pyramid %s)" % (S, args.pyramid))
            sigma = 2.0
            if getattr(args, "sigma", 0.0) > 0:
                sigma = args.sigma
            centres = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w,
                                                             indexing="ij"), -1).reshape(-1, 2).flip(-1)
                                 for h, w in shapes.tolist()], 0)                       # [S, 2] as (x, y)
            ref = centres[None, :, None, None, None, :].expand(G, q, 1, 1, 1, 2)
        wh = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
        loc_c = ref + torch.randn(G, q, M, L, P, 2, generator=g) * sigma / wh[None, None, None, :, None, :]
        loc_t = ref + torch.randn(G, q, M, W * L, P, 2, generator=g) * sigma / wh.repeat(W, 1)[None, None, None, :, None, :]
    aw = torch.softmax(torch.randn(G, q, M, L * P + W * L * P, generator=g), -1)
    aw_c = aw[..., :L * P].reshape(G, q, M, L, P)
    aw_t = aw[..., L * P:].reshape(G, q, M, W * L, P)
    grad_out = torch.randn(G, q, M * D, generator=g)
    ftab = torch.tensor([[f for f in range(T) if f != t] for t in range(T)], dtype=torch.int32)
    dev = lambda x: x.to(device=device, dtype=dtype).contiguous()
    ldt = torch.float32 if (getattr(args, "sampling", "storage") == "fp32" and dtype in (torch.bfloat16, torch.float16)) else dtype
    sam = lambda x: x.to(device=device, dtype=ldt).contiguous()
    return dict(value=dev(value), shapes=shapes.to(device), ftab=ftab.to(device),
                lsi=torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1])).to(device),
                loc_c=sam(loc_c), aw_c=sam(aw_c), loc_t=sam(loc_t), aw_t=sam(aw_t),
                grad_out=dev(grad_out), dims=(T, q, M, D, L, P, W, S))


ROUTE_KERNELS = [   # msda_last_route() phrase -> kernel symbol
    ("resident-window kernel, grad_loc/grad_attn", "msda_bwd_win_kernel"), ("forward (resident-window kernel", "msda_fwd_win_kernel"),
    ("resident-slab kernel, grad_loc/grad_attn", "msda_bwd_rs_kernel"), ("forward (resident-slab kernel", "msda_fwd_rs_kernel"),
    ("forward (slab kernel)", "msda_fwd_slab_kernel"),
    ("forward (tile kernel)", "msda_fwd_tile_kernel"), ("forward (generic kernel)", "msda_fwd_generic_kernel"),
    ("slab kernel, grad_loc/grad_attn", "msda_bwd_slab_kernel"), ("tile kernel, grad_loc/grad_attn", "msda_bwd_tile_kernel"),
    ("matrix-pipe scatter", "msda_bwd_value_mfma_kernel"),
    ("group-granular", "msda_bwd_value_grp_kernel"), ("owner-computes scatter", "msda_bwd_value_own_kernel"),
    ("per-point culling", "msda_bwd_value_points_kernel"),
    ("LDS scatter kernel)", "msda_bwd_value_lds_kernel"), ("global atomics", "msda_bwd_tile_kernel<atomics>"),
    ("backward (generic kernel)", "msda_bwd_generic_kernel"),
]


def kernel_name(route, what):
    """Kernel symbol of the forward / gather pass / scatter in a msda_last_route() string."""
    parts = [x.strip() for x in route.split(";")]
    pick = {"forward": [x for x in parts if "forward" in x],
            "gather": [x for x in parts if "grad_loc/grad_attn" in x or "global atomics" in x or "generic" in x],
            "scatter": [x for x in parts if "scatter" in x]}[what]
    for x in pick:
        for phrase, name in ROUTE_KERNELS:
            if phrase in x:
                return name
    return "msda_%s(%s)" % (what, route)


def algorithmic_bytes(args, e, gv_bytes=4, le=None):
    """Per launch over ONE clip (DESIGN.md 'algorithmic bytes'): every tensor a kernel
    must touch counted once -- value read once (not once per gathered corner), (x, y, weight) per
    sampling point, one row per query.  The backward is two kernels: the gather pass reads value,
    grad_out, loc/attn and writes grad_loc/grad_attn; the scatter pass reads loc/attn and grad_out and
    writes grad_value once (fp32, or the 16-bit storage type where the library writes it directly; accumulation happens
    in registers / LDS, so there is no read-modify-write traffic)."""
    T, q, M, D, P = args.frames, args.queries, 8, 32, 4
    shapes = PYRAMIDS[args.pyramid]
    L, W = len(shapes), T - 1
    S = sum(h * w for h, w in shapes)
    C = M * D
    points = T * q * M * (L * P + W * L * P)
    le = le or e        # bytes per sampling-location / attention-weight element (4 beside a 16-bit value with --sampling fp32)
    out = {"fwd": T * S * C * e + points * 3 * le + T * q * C * e,
           "bwd_gather": T * S * C * e + T * q * C * e + points * 3 * le + points * 3 * le,
           "bwd_scatter": points * 3 * le + T * q * C * e + T * S * C * gv_bytes}
    # the scatter as TWO kernels (round 6): the owner-computes kernel on the leading levels, the matrix-pipe kernel on the `coarse`
    # trailing ones -- each reads the points of its own levels and every grad_out row, and writes its own levels' pixels
    for coarse in (1, 2):
        px = sum(h * w for h, w in shapes[L - coarse:])
        pts = points * coarse // L
        out["bwd_scatter_mfma_%d" % coarse] = pts * 3 * le + T * q * C * e + T * px * C * gv_bytes
        out["bwd_scatter_owner_%d" % coarse] = (points - pts) * 3 * le + T * q * C * e + T * (S - px) * C * gv_bytes
    return out


def _event_ms(fn, reps, warm=3):
    """Median duration of fn() in ms: HIP events on torch's current stream (the stream the library launches on)."""
    st = torch.cuda.current_stream()
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s0, e0 in ev:
        s0.record(st)
        fn()
        e0.record(st)
    torch.cuda.synchronize()
    ts = sorted(s0.elapsed_time(e0) for s0, e0 in ev)
    return ts[len(ts) // 2]      # median: the small calls are host-bound and their mean is at the mercy of the CPU


def _plain_op_case(device, dtype, shapes, N, Lq, locs, seed, M=8, D=32, P=4):
    """Synthetic inputs of one plain MSDeformAttnFunction call.  locs: 'uniform' or 'local' (query i sits on pixel i of
    the pyramid, Lq = S, and samples N(0, (2 px)^2) around it: what an encoder layer does)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sh = torch.tensor(shapes, dtype=torch.int64)
    L, S = sh.shape[0], int(sh.prod(1).sum())
    value = torch.rand(N, S, M, D, generator=g) * 2 - 1
    if locs == "local":
        assert Lq == S
        centres = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w,
                                                         indexing="ij"), -1).reshape(-1, 2).flip(-1) for h, w in shapes], 0)
        wh = torch.stack([sh[:, 1], sh[:, 0]], -1).float()
        loc = centres[None, :, None, None, None, :] + torch.randn(N, Lq, M, L, P, 2, generator=g) * 2.0 / wh[None, None, None, :, None, :]
    else:
        loc = torch.rand(N, Lq, M, L, P, 2, generator=g)
    aw = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).reshape(N, Lq, M, L, P)
    go = torch.randn(N, Lq, M * D, generator=g)
    dev = lambda x: x.to(device=device, dtype=dtype).contiguous()
    return dict(value=dev(value), shapes=sh.to(device), lsi=torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1])).to(device),
                loc=dev(loc), aw=dev(aw), grad_out=dev(go), S=S, L=L)


def other_configs(args, device):
    """The BASELINE.json configurations beside the headline, as extra keys of the one JSON line (a few steps each;
    never part of `value`): the single-clip latency of the headline call (what DeVIS issues: 1 clip per GPU,
    main.py:85), clustered sampling locations, configs[1] (single-frame encoder attention, 800x1333, bf16) and
    configs[4] (SwinL 480x768 pyramid, fp16, im2col_step 1 vs 64; 3-level mask-head-like call)."""
    from devis_amd import _native
    from devis_amd.functions import MSDeformAttnFunction, MSDeformAttnTemporalFunction
    res = {}

    def fused_case(clips, locs, dtype, sampling="storage"):
        class A:
            pass
        a = A()
        a.clips, a.frames, a.queries, a.pyramid, a.locs = clips, args.frames, args.queries, args.pyramid, locs
        a.sampling = sampling
        b = make_clip_batch(a, device, dtype, seed=4321)
        leaves = [b[k].requires_grad_(True) for k in ("value", "loc_c", "aw_c", "loc_t", "aw_t")]

        def step():
            out = MSDeformAttnTemporalFunction.apply(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"],
                                                     b["loc_t"], b["aw_t"], clips)
            torch.autograd.grad(out, leaves, b["grad_out"])

        def fwd():
            with torch.no_grad():
                MSDeformAttnTemporalFunction.apply(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"], b["aw_c"],
                                                   b["loc_t"], b["aw_t"], clips)
        return step, fwd, clips * args.frames * args.queries

    # (i) one clip: the call DeVIS makes per decoder layer
    step, fwd, rows = fused_case(1, "uniform", torch.float32)
    ms, fms = _event_ms(step, 30, 10), _event_ms(fwd, 30, 10)
    res["single_clip_latency"] = {"workload": "cfg3, ONE clip (T=%d x %d queries), fused call, f32" % (args.frames, args.queries),
                                  "fwd_bwd_ms": round(ms, 4), "fwd_ms": round(fms, 4), "M_queries_per_s": round(rows / ms / 1e3, 3)}
    # (i-b) the same call captured in a HIP graph and replayed: the library only enqueues work (no allocation of its own, no
    # synchronisation), so forward + backward of a clip is capturable; replay shows the GPU-side floor of the call
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        gms = _event_ms(graph.replay, 30, 10)
        res["single_clip_graph"] = {"workload": "the same call (fwd + bwd of ONE clip) captured in a HIP graph, replayed",
                                    "fwd_bwd_ms": round(gms, 4), "M_queries_per_s": round(rows / gms / 1e3, 3)}
        del graph
    except Exception as exc:       # (reported, never fatal for the headline)
        res["single_clip_graph"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    # (i-b'') the same call at the query count of DeVIS's shipped YouTube-VIS configs (60 per frame; 300 is the code's default and
    # BASELINE's figure): forward + backward of one clip, eager and replayed from a HIP graph
    saved_queries = args.queries
    try:
        args.queries = 60
        step60, fwd60, rows60 = fused_case(1, "uniform", torch.float32)
        entry = {"workload": "ONE clip, T=%d x 60 queries per frame (the shipped YouTube-VIS configs), fused call, f32" % args.frames,
                 "fwd_bwd_ms": round(_event_ms(step60, 30, 10), 4), "fwd_ms": round(_event_ms(fwd60, 30, 10), 4)}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step60()
        torch.cuda.current_stream().wait_stream(side)
        graph60 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph60):
            step60()
        entry["graph_fwd_bwd_ms"] = round(_event_ms(graph60.replay, 30, 10), 4)
        entry["M_queries_per_s_graph"] = round(rows60 / entry["graph_fwd_bwd_ms"] / 1e3, 3)
        res["single_clip_60_queries"] = entry
        del graph60, step60, fwd60
    except Exception as exc:
        res["single_clip_60_queries"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    finally:
        args.queries = saved_queries
    # (i-b') the same at MODULE level: TemporalMSDeformAttnDecoder of one clip, forward + backward incl. its Linears -- eager with
    # the fused single launch, eager in the reference's call pattern (2*T operator calls + T value gathers per layer,
    # ms_deform_attn.py:325-364), and captured with torch.cuda.make_graphed_callables
    if args.pyramid == "A":
        try:
            from devis_amd.modules import TemporalMSDeformAttnDecoder
            T_, q_, C_ = args.frames, args.queries, 256
            gm = torch.Generator(device="cpu").manual_seed(11)
            shp = torch.tensor(PYRAMIDS[args.pyramid], dtype=torch.int64, device=device)
            lsi_ = torch.cat((shp.new_zeros(1), shp.prod(1).cumsum(0)[:-1]))
            S_ = int(shp.prod(1).sum())
            tsh = shp.repeat(T_ - 1, 1)
            tlsi = torch.cat((tsh.new_zeros(1), tsh.prod(1).cumsum(0)[:-1]))
            offs = [torch.tensor([t for t in range(-f, T_ - f) if t != 0], device=device) for f in range(T_)]
            mod = TemporalMSDeformAttnDecoder(T_, C_, 4, T_ - 1, 8, 4, 4).to(device)
            with torch.no_grad():
                for prm in mod.parameters():
                    prm.copy_(torch.randn(prm.shape, generator=gm).to(device) * 0.05)
            qry = torch.randn(1, T_ * q_, C_, generator=gm).to(device).requires_grad_(True)
            refp = (torch.rand(1, T_ * q_, 4, 2, generator=gm) * 0.8 + 0.1).to(device)
            srcm = torch.randn(T_, S_, C_, generator=gm).to(device).requires_grad_(True)
            wgt = torch.randn(1, T_ * q_, C_, generator=gm).to(device)
            call = lambda a, b, c: mod(a, b, c, (shp, tsh), (lsi_, tlsi), offs)[0]

            def mstep(fn=call):
                torch.autograd.grad((fn(qry, refp, srcm) * wgt).sum(), (qry, srcm))
            entry = {"workload": "TemporalMSDeformAttnDecoder (C=256, M=8, L=4, K=4) of ONE clip, T=%d x %d queries, forward + backward "
                                 "incl. value_proj / query-side Linears / output_proj, f32" % (T_, q_)}
            entry["eager_fused_ms"] = round(_event_ms(mstep, 20, 5), 4)
            mod.fused = False
            entry["eager_reference_call_pattern_ms"] = round(_event_ms(mstep, 10, 3), 4)
            mod.fused = True
            graphed = torch.cuda.make_graphed_callables(call, (qry, refp, srcm))       # gradients of the two inputs, as the eager lines
            entry["graphed_ms"] = round(_event_ms(lambda: mstep(graphed), 20, 5), 4)
            # ... and through devis_amd.graphed (round 5): the module's own signature, PARAMETER gradients included (a training step),
            # on a non-default stream as training steps through the helper must be (devis_amd/graphs.py: on the default stream the
            # helper runs the module eagerly)
            import devis_amd
            layer = devis_amd.graphed(mod, (qry, refp, srcm, (shp, tsh), (lsi_, tlsi), offs))
            helper = lambda a, b, c: layer(a, b, c, (shp, tsh), (lsi_, tlsi), offs)[0]
            with devis_amd.graph_stream(device):
                entry["graphed_with_parameter_gradients_ms"] = round(_event_ms(lambda: mstep(helper), 20, 5), 4)
            assert layer.eager_calls == 0
            # ... and the same layer fed the way DeVIS's own stack feeds it once `devis_amd.patch_transformer(dt, dvt)` is in place
            # (SURVEY 8 row f-4): the pyramid from the feature maps' Python sizes on every call (interned: the same device pair, its host
            # values known to the binding), each frame's temporal offsets through the module's `torch.tensor` (interned), the
            # repeated temporal shapes and start indices rebuilt on the device every call (graph inputs) -- one graph, no host sync
            from devis_amd import argument_builders as _ab
            itorch = _ab._InterningTorch(torch)

            def wired(a, b, c):
                shp_i, lsi_i = _ab.interned_pyramid(PYRAMIDS[args.pyramid], device)
                offs_i = [itorch.tensor([t for t in range(-f, T_ - f) if t != 0], device=device) for f in range(T_)]
                tsh_i = shp_i.repeat(T_ - 1, 1)
                tlsi_i = torch.cat((tsh_i.new_zeros((1,)), tsh_i.prod(1).cumsum(0)[:-1]))
                return layer(a, b, c, (shp_i, tsh_i), (lsi_i, tlsi_i), offs_i)[0]
            with devis_amd.graph_stream(device):
                mstep(wired)
                before = layer.graphs
                entry["graphed_from_patched_stack_wiring_ms"] = round(_event_ms(lambda: mstep(wired), 20, 5), 4)
                entry["graphs_captured_by_that_wiring"] = layer.graphs - before + 1

            def mstep_params():
                torch.autograd.grad((call(qry, refp, srcm) * wgt).sum(), [qry, srcm] + list(mod.parameters()))
            entry["eager_with_parameter_gradients_ms"] = round(_event_ms(mstep_params, 20, 5), 4)
            res["decoder_layer_module"] = entry
            del graphed, layer, helper, mod
        except Exception as exc:
            res["decoder_layer_module"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        torch.cuda.empty_cache()
    # (i-c) ONE encoder-layer call of DeVIS (the dominant cost of the transformer): the queries are the clip's
    # own pixels (Lq = S per frame, T*S = 28 920 rows -- the point count of the 16-clip decoder batch), connect-all window,
    # each query sampling around its own position in every frame (devis_transformer.py:94-121, ms_deform_attn.py:435-460)
    if args.pyramid == "A":
        saved_q = args.queries
        args.queries = sum(h * w for h, w in PYRAMIDS[args.pyramid])
        step_e, fwd_e, rows_e = fused_case(1, "local", torch.float32)
        args.queries = saved_q
        ms_e, fms_e = _event_ms(step_e, 10, 3), _event_ms(fwd_e, 10, 3)
        res["temporal_encoder_layer"] = {"workload": "TemporalMSDeformAttnEncoder call of one clip: T=%d, Lq = S = %d per frame, connect-all, local "
                                                     "sampling (N(0, (2 px)^2) around the query's own pixel), f32" % (args.frames, rows_e // args.frames),
                                         "fwd_bwd_ms": round(ms_e, 4), "fwd_ms": round(fms_e, 4), "M_queries_per_s": round(rows_e / ms_e / 1e3, 3)}
        del step_e, fwd_e
        torch.cuda.empty_cache()
    # (ii) the headline batch with clustered locations (what a trained decoder produces)
    step, fwd, rows = fused_case(args.clips, "clustered", torch.float32)
    ms = _event_ms(step, 10)
    res["clustered_locations"] = {"workload": "headline batch, reference point + N(0, (3 px)^2) offsets", "fwd_bwd_ms": round(ms, 4),
                                  "M_queries_per_s": round(rows / ms / 1e3, 3)}
    # (ii-b) a larger batch of clips (Bc up to 64): the same kernels, more work per launch
    step, fwd, rows = fused_case(64, "uniform", torch.float32)
    ms = _event_ms(step, 6)
    res["batch_64_clips"] = {"workload": "cfg3, 64 clips per step (headline: %d), uniform locations, f32" % args.clips,
                             "fwd_bwd_ms": round(ms, 4), "M_queries_per_s": round(rows / ms / 1e3, 3)}
    del step, fwd
    torch.cuda.empty_cache()
    # (ii-c) the decoder batch on the 800x1333 pyramid (BASELINE configs[1]'s image size with the decoder's query count): two levels
    # stay outside the resident slab, so route rules tuned on the 360x640 pyramid are checked against this one too
    if args.pyramid == "A":
        args.pyramid = "B"
        try:
            for key, dt in (("decoder_800x1333_f32", torch.float32), ("decoder_800x1333_bf16", torch.bfloat16)):
                step, fwd, rows = fused_case(8, "uniform", dt)
                ms, fms = _event_ms(step, 10, 5), _event_ms(fwd, 10, 5)
                res[key] = {"workload": "cfg3 decoder call on the 800x1333 pyramid (S = 22223), 8 clips, uniform locations, %s" % str(dt)[6:],
                            "fwd_bwd_ms": round(ms, 4), "fwd_ms": round(fms, 4), "M_queries_per_s": round(rows / ms / 1e3, 3)}
                del step, fwd
                torch.cuda.empty_cache()
        finally:
            args.pyramid = "A"
    # (iii) the headline batch in the 16-bit storage types (arithmetic stays fp32)
    for key, dt, sampling in (("headline_bf16", torch.bfloat16, "storage"), ("headline_f16", torch.float16, "storage"),
                              ("headline_bf16_fp32_sampling", torch.bfloat16, "fp32")):
        step, fwd, rows = fused_case(args.clips, "uniform", dt, sampling)
        ms, fms = _event_ms(step, 10), _event_ms(fwd, 10)
        res[key] = {"workload": "headline batch, %s storage%s" % (key.split("_")[1], ", float32 sampling locations / attention weights "
                                                                  "(MSDA_BF16_LOC32, the modules' default)" if sampling == "fp32" else ""), "fwd_bwd_ms": round(ms, 4), "fwd_ms": round(fms, 4),
                    "M_queries_per_s": round(rows / ms / 1e3, 3)}
    del step, fwd

    def plain(name, dtype, shapes, N, Lq, locs, steps, reps=6, per_kernel=False):
        c = _plain_op_case(device, dtype, shapes, N, Lq, locs, seed=99)
        leaves = [c[k].requires_grad_(True) for k in ("value", "loc", "aw")]
        out_rows = N * Lq
        entry = {"workload": name}
        for st in steps:
            def step():
                out = MSDeformAttnFunction.apply(c["value"], c["shapes"], c["lsi"], c["loc"], c["aw"], st)
                torch.autograd.grad(out, leaves, c["grad_out"])

            def fwd():
                with torch.no_grad():
                    MSDeformAttnFunction.apply(c["value"], c["shapes"], c["lsi"], c["loc"], c["aw"], st)
            ms, fms = _event_ms(step, reps), _event_ms(fwd, reps)
            entry["im2col_step_%d" % st] = {"fwd_ms": round(fms, 4), "fwd_bwd_ms": round(ms, 4),
                                            "fwd_M_queries_per_s": round(out_rows / fms / 1e3, 2),
                                            "fwd_bwd_M_queries_per_s": round(out_rows / ms / 1e3, 2)}
        if per_kernel:      # per-kernel durations and algorithmic-byte fractions of the un-chunked call
            e = c["value"].element_size()
            M, D, P, L, S = 8, 32, 4, c["L"], c["S"]
            pts = N * Lq * M * L * P
            alg = {"fwd": N * S * M * D * e + pts * 3 * e + N * Lq * M * D * e,
                   "gather": N * S * M * D * e + N * Lq * M * D * e + pts * 6 * e,
                   "scatter": pts * 3 * e + N * Lq * M * D * e + N * S * M * D * (4 if _native.grad_value_dtype(leaves[0], c["shapes"], Lq, L, P) == torch.float32 else e)}
            dv = [x.detach() for x in leaves]
            out = torch.empty((N, Lq, M * D), dtype=dtype, device=device)
            gv = torch.empty(c["value"].shape, dtype=_native.grad_value_dtype(dv[0], c["shapes"], Lq, L, P), device=device)
            gl, ga = torch.empty_like(c["loc"]), torch.empty_like(c["aw"])
            ws = _native.bwd_workspace(device, N, Lq, M, L)
            t = {"fwd": _event_ms(lambda: _native.forward(dv[0], c["shapes"], c["lsi"], dv[1], dv[2], out), reps)}
            names = {"fwd": kernel_name(_native.last_route(), "forward")}
            os.environ["MSDA_ENABLE_HOOKS"] = "1"
            for ph, key, what in (("1", "gather", "gather"), ("2", "scatter", "scatter")):
                os.environ["MSDA_BWD_PHASES"] = ph
                _native.reload_knobs()

                def bwd():
                    ws[:16].zero_()
                    load = _native.load()
                    rc = load.msda_backward(_native.dtype_code(dtype), dv[0].data_ptr(), c["shapes"].data_ptr(), c["lsi"].data_ptr(),
                                            dv[1].data_ptr(), dv[2].data_ptr(), c["grad_out"].data_ptr(), N, S, M, D, L, Lq, P,
                                            gv.data_ptr(), _native.dtype_code(gv.dtype), gl.data_ptr(), ga.data_ptr(), ws.data_ptr(), ws.numel() * 4, None,
                                            _native.shapes_hint(c["shapes"]), torch.cuda.current_stream().cuda_stream)
                    assert rc == 0, load.msda_last_error()
                t[key] = _event_ms(bwd, reps)
                names[key] = kernel_name(_native.last_route(), what)
            os.environ.pop("MSDA_BWD_PHASES")
            os.environ.pop("MSDA_ENABLE_HOOKS")
            _native.reload_knobs()
            entry["kernels"] = {names[k]: {"avg_ms": round(t[k], 4), "algorithmic_GBps": round(alg[k] / t[k] / 1e6, 1),
                                           "frac_of_hbm_peak": round(alg[k] / t[k] / 1e6 / HBM_PEAK_GBS, 4)} for k in t}
        res_key = name.split(":")[0]
        res[res_key] = entry
        del c, leaves
        torch.cuda.empty_cache()

    # (v) the fused temporal ENCODER call at 800x1333 (what dominates a DeVIS step: Lq = S = 22223 per frame, T = 6, 102 M
    # sampling points per layer), fp32, per kernel
    def temporal_encoder_b():
        class A:
            pass
        a = A()
        a.clips, a.frames, a.queries, a.pyramid, a.locs, a.sampling = 1, args.frames, 22223, "B", "local", "storage"
        b = make_clip_batch(a, device, torch.float32, seed=777)
        T, q, M, D, L, P, W, S = b["dims"]
        dv = [b[k] for k in ("value", "loc_c", "aw_c", "loc_t", "aw_t")]
        out = torch.empty((T, q, M * D), dtype=torch.float32, device=device)
        gv = torch.zeros(b["value"].shape, dtype=torch.float32, device=device)
        grads = [torch.empty_like(x) for x in dv[1:]]
        ws = _native.bwd_workspace(device, T, q, M, L * (1 + W))
        t, names = {}, {}
        t["fwd"] = _event_ms(lambda: _native.temporal_forward(dv[0], b["shapes"], b["lsi"], b["ftab"], dv[1], dv[2], dv[3], dv[4], 1, out), 5, 2)
        names["fwd"] = kernel_name(_native.last_route(), "forward")
        def bwd():
            ws[:16].zero_()
            _native.temporal_backward(dv[0], b["shapes"], b["lsi"], b["ftab"], dv[1], dv[2], dv[3], dv[4], b["grad_out"], 1, gv, *grads, workspace=ws)
        os.environ["MSDA_ENABLE_HOOKS"] = "1"
        for ph, key in (("1", "gather"), ("2", "scatter")):
            os.environ["MSDA_BWD_PHASES"] = ph
            _native.reload_knobs()
            t[key] = _event_ms(bwd, 5, 2)
            names[key] = kernel_name(_native.last_route(), key)
        os.environ.pop("MSDA_BWD_PHASES"); os.environ.pop("MSDA_ENABLE_HOOKS")
        _native.reload_knobs()
        class B:
            pass
        ab_args = B(); ab_args.frames, ab_args.queries, ab_args.pyramid = T, q, "B"
        ab = algorithmic_bytes(ab_args, 4, 4)
        alg = {"fwd": ab["fwd"], "gather": ab["bwd_gather"], "scatter": ab["bwd_scatter"]}
        total = sum(t.values())
        return {"workload": "fused temporal encoder call at 800x1333: ONE clip, T=%d, Lq = S = %d per frame, L=4, K=4, M=8xD=32, fp32, local "
                            "sampling (N(0, (2 px)^2) round the query's own pixel)" % (T, q),
                "fwd_bwd_ms": round(total, 4), "M_queries_per_s": round(T * q / total / 1e3, 3),
                "kernels": {names[k]: {"avg_ms": round(t[k], 4), "algorithmic_GBps": round(alg[k] / t[k] / 1e6, 1),
                                       "frac_of_hbm_peak": round(alg[k] / t[k] / 1e6 / HBM_PEAK_GBS, 4)} for k in t}}
    res["temporal_encoder_800x1333_f32"] = temporal_encoder_b()
    torch.cuda.empty_cache()

    plain("cfg1_encoder_800x1333_bf16: single-frame encoder attention, N=8 images, Lq = S = 22223, L=4, K=4, M=8xD=32, bf16, "
          "encoder-like local sampling", torch.bfloat16, PYRAMIDS["B"], 8, 22223, "local", (64,), per_kernel=True)
    swin = [(60, 96), (30, 48), (15, 24), (8, 12)]            # SwinL training size 480x768 (src/datasets/vis.py:228-231)
    plain("cfg4_swinl_fp16_decoder_like: plain MSDeformAttn, N = T = 6 frames as the batch, 300 queries, SwinL 480x768 pyramid, "
          "fp16, im2col_step 1 vs 64", torch.float16, swin, 6, 300, "uniform", (1, 64))
    plain("cfg4_swinl_fp16_encoder_like: same pyramid, Lq = S = 7656 (local sampling), fp16, im2col_step 1 vs 64",
          torch.float16, swin, 6, 7656, "local", (1, 64))
    plain("cfg4_mask_head_like_fp16: 3 levels (/8, /16, /32), Lq = 10 instances x 6 frames, N=1 (the reference mask head "
          "itself holds no MSDeformAttn: deformable_segmentation.py:265 is torchvision deform_conv2d)",
          torch.float16, swin[:3], 1, 60, "uniform", (64,))
    return res


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, script=None, check_devices=True, timeout=None):
    """`python bench.py --gpus N` started WITHOUT a launcher: run the N ranks as a CHILD `python -m torch.distributed.run`
    (one process per GPU over RCCL, the reference's own launch: main.py:129-143, src/util/misc.py:437-460), relay rank 0's one
    JSON line and return the child's exit status.  Called before anything in this process has initialised the GPU
    (`torch.cuda.device_count()` does not, on this image) and never through `os.exec*`.  Fewer than N visible devices is an
    error, never a silent `n_gpus: 1` line.  `script` / `check_devices`: the CPU test drives a stub worker through the same code."""
    import subprocess
    if check_devices:
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d asked for, %d device(s) visible -- refusing to run fewer ranks\n" % (n, have))
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL needs it)
    env["BENCH_LAUNCHED_RANKS"] = str(n)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
        sys.stderr.write("bench.py: the %d-rank child did not finish in %s s\n" % (n, timeout))
        return 124
    line = None
    for text in out.splitlines():
        text = text.strip()
        if text.startswith("{") and text.endswith("}"):
            try:
                json.loads(text)
                line = text
            except ValueError:
                pass
    rc = proc.returncode
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the %d-rank child exited 0 without a JSON line\n" % n)
        rc = 1
    if rc == 0:
        got = json.loads(line)
        if got.get("n_gpus") != n or got.get("ranks_seen", n) != n:
            sys.stderr.write("bench.py: asked for %d ranks, the line reports n_gpus=%s ranks_seen=%s\n"
                             % (n, got.get("n_gpus"), got.get("ranks_seen")))
            rc = 1
    if rc != 0:
        sys.stderr.write("bench.py: %d-rank run failed (status %d)\n" % (n, rc))
        return rc
    print(line, flush=True)
    return 0


def main():
    args = parse_args()
    if args.share_device and args.dist_backend != "gloo":
        sys.exit("bench.py: --share-device needs --dist-backend gloo (RCCL refuses two ranks on one device)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: be the launcher (a child process; this one has not touched the GPU)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], check_devices=not args.share_device))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # a launcher started another number of ranks than --gpus names: the line would lie about N either way
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or drop the launcher and let "
                 "`python bench.py --gpus %d` start its own ranks)" % (args.gpus, world, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.share_device:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        sys.exit("bench.py: rank %d has no device (%d visible)" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # tensors of the few scalar reductions of the protocol: on the device for RCCL, on the host for gloo
    red_dev = device if args.dist_backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)   # RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    elif args.mode == "sharded":                            # single process: degenerate collectives
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)

    from devis_amd.functions import MSDeformAttnFunction, MSDeformAttnTemporalFunction
    dtype = DTYPES[args.dtype]
    b = make_clip_batch(args, device, dtype, seed=1234 + rank)
    T, q, M, D, L, P, W, S = b["dims"]
    if args.value_layout == "padded":
        buf = torch.zeros((b["value"].shape[0], S, M + 1, D), dtype=dtype, device=device)
        buf[:, :, :M] = b["value"]
        b["value"] = buf[:, :, :M]
    leaves = [b[k].requires_grad_(True) for k in ("value", "loc_c", "aw_c", "loc_t", "aw_t")]
    t_shapes, t_lsi = None, None
    if args.pattern == "reference":
        t_shapes = b["shapes"].repeat(W, 1)
        t_lsi = torch.cat((t_shapes.new_zeros(1), t_shapes.prod(1).cumsum(0)[:-1]))

    def forward():
        if args.pattern == "fused":
            return MSDeformAttnTemporalFunction.apply(b["value"], b["shapes"], b["lsi"], b["ftab"], b["loc_c"],
                                                      b["aw_c"], b["loc_t"], b["aw_t"], args.clips)
        outs = []   # the reference's per-frame loop (ms_deform_attn.py:325-364), clip by clip
        for g in range(args.clips * T):
            c, t = divmod(g, T)
            o1 = MSDeformAttnFunction.apply(b["value"][g][None], b["shapes"], b["lsi"], b["loc_c"][g][None],
                                            b["aw_c"][g][None], 64)
            frames = b["ftab"][t].long() + c * T
            stacked = b["value"][frames].flatten(0, 1)[None]
            o2 = MSDeformAttnFunction.apply(stacked, t_shapes, t_lsi, b["loc_t"][g][None], b["aw_t"][g][None], 64)
            outs.append(o1 + o2)
        return torch.cat(outs, 0)

    def step():
        out = forward()
        torch.autograd.grad(out, leaves, b["grad_out"])

    if args.mode == "sharded":
        # Mode 2: the SAME `clips` clips on every rank (seed without rank), each split over the ranks:
        # value rows chunked over the flattened T*S axis, queries chunked per frame.
        from devis_amd import clip_parallel as cp
        b = make_clip_batch(args, device, dtype, seed=1234)
        rows = T * S
        chunk = cp.padded_chunk(rows, world)
        q0, q1 = cp.shard_range(q, world, rank)
        shard = []
        for c in range(args.clips):
            flat = b["value"][c * T:(c + 1) * T].reshape(rows, M, D)
            pad = torch.zeros((chunk * world, M, D), dtype=dtype, device=device)
            pad[:rows] = flat
            cut = lambda k: b[k][c * T:(c + 1) * T, q0:q1].contiguous().requires_grad_(True)
            shard.append(dict(v=pad[rank * chunk:(rank + 1) * chunk].clone().requires_grad_(True),
                              lc=cut("loc_c"), ac=cut("aw_c"), lt=cut("loc_t"), at=cut("aw_t"),
                              go=b["grad_out"][c * T:(c + 1) * T, q0:q1].contiguous()))

        transport = {"storage": None, "bf16": torch.bfloat16, "f16": torch.float16}[args.transport]
        sh_leaves = [x for sh in shard for x in (sh["v"], sh["lc"], sh["ac"], sh["lt"], sh["at"])]
        sh_go = [sh["go"] for sh in shard]

        def step():  # noqa: F811
            # every clip's all-gather is issued before the first kernel (the collectives of the later clips travel while the
            # kernels of the earlier ones run), one backward over all clips (their reduce-scatters follow each other on the
            # communication stream while the next clip's kernels run)
            outs = cp.sharded_temporal_attention_batch([(sh["v"], sh["lc"], sh["ac"], sh["lt"], sh["at"]) for sh in shard],
                                                       T, S, b["shapes"], b["lsi"], b["ftab"], transport_dtype=transport)
            torch.autograd.grad(outs, sh_leaves, sh_go)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # Clock / allocator / code-object warm-up before the contract's W warm-up steps: a fresh box needs a few hundred
    # milliseconds of load before its clocks settle (measured: 1.83 ms per step in the first 34 ms after start-up
    # against 1.68 ms later with identical per-kernel times), and K may be small.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_seconds:
        step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    ms_per_step = elapsed / args.steps * 1e3
    rows_per_step = (1 if args.mode == "sharded" else world) * args.clips * T * q
    value = rows_per_step / (ms_per_step * 1e-3) / 1e6

    # ---- the same K steps on the layout devis_amd's own modules hand the op (value_proj writes one spare
    # head slot per pixel row, functions.project_value): reported beside the headline, never as `value`
    padded_line = None
    if args.value_layout == "dense" and args.pattern == "fused" and args.mode == "clip-parallel":
        # INTERLEAVED blocks (A-B-A-B, medians) so that clock / thermal drift inside the run cannot favour either layout:
        # dense = what `value` above reports (the reference's tensor), padded = what devis_amd's value_proj writes
        dense_value = b["value"]
        buf = torch.zeros((dense_value.shape[0], S, M + 1, D), dtype=dtype, device=device)
        buf[:, :, :M] = dense_value.detach()
        padded_value = buf[:, :, :M].requires_grad_(True)
        block = max(8, args.steps // 8)          # (>= 8 steps per block whatever --steps: 2-step blocks measured noise)

        def timed_block(v):
            b["value"] = v
            leaves[0] = v
            step(); step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(block):
                step()
            barrier()
            return (time.perf_counter() - t0) / block

        times = {"dense": [], "padded": []}
        for _ in range(4):
            times["dense"].append(timed_block(dense_value))
            times["padded"].append(timed_block(padded_value))
        med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([med["dense"], med["padded"]], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            med = {"dense": tt[0].item(), "padded": tt[1].item()}
        padded_line = {"layout": "value rows padded by one head slot (what devis_amd's value_proj writes)",
                       "method": "4 interleaved blocks of %d steps per layout, medians" % block,
                       "ms_per_step": round(med["padded"] * 1e3, 4),
                       "value": round(rows_per_step / med["padded"] / 1e6, 3),
                       "dense_ms_per_step_same_method": round(med["dense"] * 1e3, 4),
                       "dense_value_same_method": round(rows_per_step / med["dense"] / 1e6, 3), "unit": "M-queries/s"}
        b["value"] = dense_value
        leaves[0] = dense_value
        del buf, padded_value

    # ---- per-kernel durations with HIP events on the launch stream (fused pattern only) ----------
    roofline, extra = None, {}
    if rank == 0 and args.pattern == "fused" and args.mode == "clip-parallel":
        from devis_amd import _native
        stream = torch.cuda.current_stream()
        out = torch.empty((args.clips * T, q, M * D), dtype=dtype, device=device)
        acc = _native.grad_value_dtype(b["value"], b["shapes"], q, L, P, clips=args.clips, window=W, Pt=P)   # fp32, or the 16-bit storage type
        gv = torch.zeros(b["value"].shape, dtype=acc, device=device)
        gl_c, ga_c = torch.empty_like(b["loc_c"]), torch.empty_like(b["aw_c"])
        gl_t, ga_t = torch.empty_like(b["loc_t"]), torch.empty_like(b["aw_t"])
        dv = [x.detach() for x in (b["value"], b["loc_c"], b["aw_c"], b["loc_t"], b["aw_t"])]

        def time_kernel(fn, reps):
            for _ in range(3):
                fn()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for s, e in ev:
                s.record(stream)
                fn()
                e.record(stream)
            torch.cuda.synchronize()
            ts = sorted(s.elapsed_time(e) for s, e in ev)
            return sum(ts) / len(ts), ts[len(ts) // 2]

        fwd_ms, fwd_med = time_kernel(lambda: _native.temporal_forward(
            dv[0], b["shapes"], b["lsi"], b["ftab"], dv[1], dv[2], dv[3], dv[4], args.clips, out), 20)
        routes = {"forward": _native.last_route()}
        fwd_name = kernel_name(routes["forward"], "forward")        # the kernels the library actually launched
        # the backward entry point launches two kernels; MSDA_BWD_PHASES lets each be timed alone
        ws = _native.bwd_workspace(device, args.clips * T, q, M, L * (1 + W))
        def bwd():
            ws[:16].zero_()     # the scatter pass takes its work tickets from the zeroed head of the workspace
            _native.temporal_backward(
                dv[0], b["shapes"], b["lsi"], b["ftab"], dv[1], dv[2], dv[3], dv[4], b["grad_out"], args.clips,
                gv, gl_c, ga_c, gl_t, ga_t, workspace=ws)
        # (a measurement hook: honoured only with MSDA_ENABLE_HOOKS=1 and re-read on request; it is switched on
        # here, AFTER the timed region, and off again below)
        os.environ["MSDA_ENABLE_HOOKS"] = "1"
        os.environ["MSDA_BWD_PHASES"] = "1"
        _native.reload_knobs()
        gat_ms, gat_med = time_kernel(bwd, 20)
        routes["gather_pass"] = _native.last_route()
        gat_name = kernel_name(routes["gather_pass"], "gather")
        os.environ["MSDA_BWD_PHASES"] = "2"
        _native.reload_knobs()
        sca_ms, sca_med = time_kernel(bwd, 20)
        routes["scatter"] = _native.last_route()
        e = b["value"].element_size()
        ab = algorithmic_bytes(args, e, gv.element_size(), b["loc_c"].element_size())
        scatter_parts = [x.strip() for x in routes["scatter"].split(";") if "scatter" in x and "zero-fill" not in x]
        kernels = {}
        if len(scatter_parts) == 2 and "matrix-pipe" in scatter_parts[1]:
            # two kernels: each timed alone (MSDA_SCATTER_PART, a measurement knob); the pair's time is reported beside them
            coarse = 2 if sum(h * w for h, w in PYRAMIDS[args.pyramid][-2:]) <= 303 else 1
            for part, key, label in ((1, "owner", " (grad_value scatter, levels [0, %d))" % (L - coarse)),
                                     (2, "mfma", " (grad_value scatter, the %d coarse level(s) on the matrix pipe)" % coarse)):
                os.environ["MSDA_SCATTER_PART"] = str(part)
                _native.reload_knobs()
                ms, med = time_kernel(bwd, 20)
                kernels[kernel_name(scatter_parts[part - 1], "scatter") + label] = (ms, med, ab["bwd_scatter_%s_%d" % (key, coarse)])
            os.environ.pop("MSDA_SCATTER_PART")
            sca_name = "both scatter kernels"
        else:
            sca_name = kernel_name(routes["scatter"], "scatter")
            kernels[sca_name + " (grad_value scatter)"] = (sca_ms, sca_med, ab["bwd_scatter"])
        os.environ.pop("MSDA_BWD_PHASES")
        os.environ.pop("MSDA_ENABLE_HOOKS")
        _native.reload_knobs()
        kernels.update({
            fwd_name: (fwd_ms, fwd_med, ab["fwd"]),
            gat_name + " (grad_loc/grad_attn gather pass)": (gat_ms, gat_med, ab["bwd_gather"]),
        })
        dom = max(kernels, key=lambda k: kernels[k][0])
        d_ms, _, d_bytes = kernels[dom]
        ach = d_bytes * args.clips / (d_ms * 1e-3) / 1e9
        # HBM traffic cannot be counted from inside this process (PMC counters need rocprofv3 around it); quote the committed
        # rocprofv3 result (profiles/hbm_traffic.json) when it was made on this configuration AND from the very kernel sources
        # of this build (content hash of devis_amd/csrc), else null
        traffic, traffic_src = None, None
        try:
            from devis_amd import build as _b
            prof = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            w = prof["workload"]
            if (w["clips"], w["frames"], w["queries"], w["pyramid"], w["dtype"], w["locs"], w["pattern"]) == \
                    (args.clips, args.frames, args.queries, args.pyramid, args.dtype, args.locs, args.pattern) \
                    and args.value_layout == "dense" and prof.get("source_hash") == _b._source_hash():
                k = prof["kernels"][dom.split(" ")[0]]
                traffic = k["fetch_bytes"] + k["write_bytes"]
                traffic_src = prof.get("source", "profiles/hbm_traffic.json")
        except (OSError, KeyError, ValueError):
            traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": d_bytes * args.clips, "avg_launch_ms": round(d_ms, 4),
                    "traffic_source": traffic_src}
        # What the kernels ARE bound by (none of them is HBM-bound): busy fractions of the on-chip units from the committed PMC
        # passes (profiles/onchip.json, scripts/make_onchip.py), quoted -- like `traffic` -- only for this configuration and this
        # very build of the kernel sources; the L2 request rate uses the launch durations measured above.
        onchip_all = None
        try:
            from devis_amd import build as _b
            prof = json.load(open(os.path.join(ROOT, "profiles", "onchip.json")))
            w = prof["workload"]
            if (w["clips"], w["frames"], w["queries"], w["pyramid"], w["dtype"], w["locs"], w["pattern"]) == \
                    (args.clips, args.frames, args.queries, args.pyramid, args.dtype, args.locs, args.pattern) \
                    and args.value_layout == "dense" and prof.get("source_hash") == _b._source_hash():
                onchip_all = {}
                for name, (ms, _, _) in kernels.items():
                    k = dict(prof["kernels"].get(name.split(" ")[0], {}))
                    if not k:
                        continue
                    if "l2_requests" in k:
                        k["l2_request_GBps"] = round(k["l2_requests"] * 128 / (ms * 1e-3) / 1e9, 1)
                        k["l2_request_frac_of_34.5TBps"] = round(k["l2_request_GBps"] / 34500.0, 4)
                    onchip_all[name] = k
                roofline["onchip"] = dict(onchip_all.get(dom, {}), source=prof.get("source"))
        except (OSError, KeyError, ValueError):
            onchip_all = None
        # the fraction of the NOMINAL peak above, and of what a plain device-to-device copy reaches on this very box
        # (1 GiB read + 1 GiB written per copy, HIP events, median of 10)
        try:
            src_buf = torch.empty(256 << 20, dtype=torch.float32, device=device).fill_(1.0)
            dst_buf = torch.empty_like(src_buf)
            copy_ms = _event_ms(lambda: dst_buf.copy_(src_buf), 10, 3)
            copy_gbs = 2 * src_buf.numel() * 4 / (copy_ms * 1e-3) / 1e9
            roofline["measured_copy_GBps"] = round(copy_gbs, 1)
            roofline["frac_of_measured_copy"] = round(ach / copy_gbs, 4)
            del src_buf, dst_buf
        except RuntimeError:
            pass
        extra = {"kernels": {k: {"avg_ms": round(v[0], 4), "median_ms": round(v[1], 4),
                                 "algorithmic_GBps": round(v[2] * args.clips / (v[0] * 1e-3) / 1e9, 1),
                                 "frac_of_hbm_peak": round(v[2] * args.clips / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                             for k, v in kernels.items()},
                 "scatter_pass": {"kernels": sca_name, "avg_ms": round(sca_ms, 4), "median_ms": round(sca_med, 4),
                                  "algorithmic_GBps": round(ab["bwd_scatter"] * args.clips / (sca_ms * 1e-3) / 1e9, 1),
                                  "frac_of_hbm_peak": round(ab["bwd_scatter"] * args.clips / (sca_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                 "forward_frac_of_hbm_peak_vs_north_star_0.60": round(ab["fwd"] * args.clips / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "forward_M_queries_per_s": round(args.clips * T * q / (fwd_ms * 1e-3) / 1e6, 2),
                 "onchip": onchip_all,
                 "routes": routes}       # msda_last_route() of the three launches timed above: which kernel family ran, as the library says

    # ---- CPU baseline: the reference's pure-PyTorch path on the host cores (bounded sample) -----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import msda_oracle as O
        # grid_sample's backward degrades badly when oversubscribed (256 threads: >300 s per clip);
        # 16 threads is near its best on this host -- `cores` reports what was actually used
        cores = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(cores)
        cb = {k: (v[:T].detach().float().cpu() if k in ("value", "loc_c", "aw_c", "loc_t", "aw_t", "grad_out") else v.cpu())
              for k, v in b.items() if k != "dims"}
        c_shapes_t = cb["shapes"].repeat(W, 1)

        def cpu_frame(t):
            """one frame of the reference's loop (ms_deform_attn.py:325-364): current + temporal call,
            forward and backward through autograd"""
            lv = [cb[k].clone().requires_grad_(True) for k in ("value", "loc_c", "aw_c", "loc_t", "aw_t")]
            o1 = O.grid_sample_forward(lv[0][t][None], cb["shapes"], lv[1][t][None], lv[2][t][None])
            stacked = lv[0][cb["ftab"][t].long()].flatten(0, 1)[None]
            o2 = O.grid_sample_forward(stacked, c_shapes_t, lv[3][t][None], lv[4][t][None])
            torch.autograd.grad(o1 + o2, lv, cb["grad_out"][t][None])

        cpu_frame(0)                                   # warm-up
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < args.cpu_seconds:
            cpu_frame(n % T)
            n += 1
        dt = time.perf_counter() - t0
        cpu = {"value": round(n * q / dt / 1e6, 6), "unit": "M-queries/s", "cores": cores, "kind": "port",
               "sample": "%d frame passes (each: current + temporal call, fwd+bwd, %d queries, pyramid %s, fp32) of "
                         "oracle.grid_sample_forward in the reference's call pattern, %.1f s, %d torch threads"
                         % (n, q, args.pyramid, dt, cores)}
        # The same frame passes with the tensors on THIS GPU: what the reference's pure-PyTorch formulation (F.grid_sample, the
        # only form of the reference's path that runs without its CUDA extension) does on an MI355X.  A second baseline beside
        # the host-core one, never `value`; a failure here must not cost the bench line.
        try:
            cb = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in cb.items()}
            c_shapes_t = c_shapes_t.to(device)
            for t in range(T):
                cpu_frame(t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(4 * T):
                cpu_frame(i % T)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            cpu["same_formulation_on_this_gpu"] = {
                "value": round(4 * T * q / dt / 1e6, 4), "unit": "M-queries/s",
                "sample": "%d frame passes of the same torch program (F.grid_sample + autograd, reference call pattern, fp32) on the "
                          "GPU, %.1f ms per clip" % (4 * T, dt / 4 * 1e3)}
        except Exception as e:      # noqa: BLE001 -- a baseline, not the product
            cpu["same_formulation_on_this_gpu"] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}

    # every rank contributes a one: the SCALE record can check that RCCL really saw N ranks
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        ones = torch.ones(1, device=red_dev, dtype=torch.int32)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(ones.item())
    others = None
    if rank == 0 and world == 1 and not args.no_other_configs and args.mode == "clip-parallel":
        del b, leaves
        torch.cuda.empty_cache()
        others = other_configs(args, device)

    if rank == 0:
        line = {
            "metric": "MSDeformAttn fwd+bwd M-queries/s at T=6,L=4,K=4,C=256", "value": round(value, 3),
            "unit": "M-queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if args.mode == "sharded" else "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "cfg3 DeVIS decoder temporal MSDeformAttn, one layer fwd+bwd: T=%d frames, "
                                   "%d queries/frame, L=4, K=4, C=256 (M=8xD=32), pyramid %s (S=%d), %d clips/GPU/step, "
                                   "%s call pattern, %s sampling locations%s, %s value layout"
                                   % (T, q, args.pyramid, S, args.clips, args.pattern, args.locs,
                                      " in float32" if (args.sampling == "fp32" and args.dtype != "f32") else "", args.value_layout),
                       "clips_per_gpu": args.clips, "query_rows_per_step": rows_per_step,
                       "parallelism": ("clip-parallel x%d (no data-path collective)" % world) if args.mode == "clip-parallel"
                       else "one clip sharded x%d (all-gather value / reduce-scatter grad_value over RCCL, all gathers of a step in "
                            "flight before its first kernel, %s transport)" % (world, args.transport)},
            "roofline": roofline, "cpu_baseline": cpu, "ranks_seen": ranks_seen,
            **({"test_run": "--share-device: all %d ranks on ONE GPU over gloo -- the N-rank protocol only, not a scaling figure" % world}
               if args.share_device else {}),
            "prewarm_seconds": args.prewarm_seconds,
        }
        line.update(extra)
        if others is not None:
            line["other_configs"] = others
        if padded_line is not None:
            line["padded_value_layout"] = padded_line
    if world > 1 or args.mode == "sharded":
        import torch.distributed as dist
        dist.destroy_process_group()            # (before the line: RCCL logs its library path when the group goes down)
        sys.stdout.flush(); sys.stderr.flush()
        if rank == 0 and world > 1:
            time.sleep(1.0)                     # (the other ranks' teardown messages first)
    if rank == 0:
        print(json.dumps(line), flush=True)     # the ONE JSON line, last thing on stdout
    if world > 1 or args.mode == "sharded":
        # RCCL logs "Librccl path : ..." when the process exits; nothing may follow the JSON line on stdout
        sys.stdout.flush()
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)


if __name__ == "__main__":
    main()
