"""Worker for the multi-process tests (spawned by test_dist_cpu.py / test_dist_gpu.py).

Checks Mode 2 (one clip sharded over the ranks, devis_amd/clip_parallel.py): outputs of the ranks
concatenated == the unsharded result, and after the reduce-scatter every rank holds exactly the
gradient of its own value chunk.  On CPU (gloo) the kernels are replaced by the oracle-backed test
double; on GPU (nccl = RCCL) the real HIP path runs."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


class _Patch:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


def main():
    backend = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        device = torch.device("cuda", torch.cuda.current_device())
        dist.init_process_group("nccl", device_id=device)
    else:
        device = torch.device("cpu")
        dist.init_process_group("gloo")
        import fake_native
        fake_native.install(_Patch())
    from helpers import make_temporal_inputs, temporal_reference
    from devis_amd import clip_parallel as cp

    case = sys.argv[2] if len(sys.argv) > 2 else "small"
    if case == "cfg3":      # BASELINE configs[3]: the T = 6 decoder clip (frames do not divide 8 ranks), connect-all window
        from helpers import PYR_A
        T, W, M, D, Lq = 6, 5, 8, 32, 300          # 300 queries per frame: the decoder clip at its full size
        d = make_temporal_inputs(9, T, W, M, D, Lq, PYR_A, 4, 4, dtype=np.float64)
    else:
        T, W, M, D, Lq = 3, 2, 4, 8, 11
        d = make_temporal_inputs(9, T, W, M, D, Lq, [(6, 5), (3, 3)], 3, 2, dtype=np.float64)
    ref = temporal_reference(*(d[k] for k in ("value", "shapes", "lsi", "ftab", "loc_c", "aw_c", "loc_t",
                                              "aw_t", "grad_out")))
    S = d["value"].shape[1]
    dt = torch.float64
    # this rank's chunk of the flattened value rows (padded to equal shards) and its query range
    rows = T * S
    chunk = cp.padded_chunk(rows, world)
    flat = np.zeros((chunk * world, M, D))
    flat[:rows] = d["value"].reshape(rows, M, D)
    v_chunk = torch.from_numpy(flat[rank * chunk:(rank + 1) * chunk]).to(device, dt).requires_grad_(True)
    q0, q1 = cp.shard_range(Lq, world, rank)
    cut = lambda k: torch.from_numpy(np.ascontiguousarray(d[k][:, q0:q1])).to(device, dt).requires_grad_(True)
    lc, ac, lt, at = cut("loc_c"), cut("aw_c"), cut("loc_t"), cut("aw_t")
    shapes = torch.from_numpy(d["shapes"]).to(device)
    lsi = torch.from_numpy(d["lsi"]).to(device)
    ftab = torch.from_numpy(d["ftab"]).to(device)

    # the backward collective must be the reduce-scatter an 8-GPU node runs (not an all-reduce stand-in): count the calls
    calls = {"rs": 0, "ar": 0, "rs_async": 0}
    real_rs, real_ar = dist.reduce_scatter_tensor, dist.all_reduce
    def counting_rs(*a, **k):
        calls["rs"] += 1
        calls["rs_async"] += 1 if k.get("async_op") else 0
        return real_rs(*a, **k)
    def counting_ar(*a, **k):
        calls["ar"] += 1
        return real_ar(*a, **k)
    dist.reduce_scatter_tensor, dist.all_reduce = counting_rs, counting_ar

    out = cp.sharded_temporal_attention(v_chunk, T, S, shapes, lsi, ftab, lc, ac, lt, at)
    go = torch.from_numpy(np.ascontiguousarray(d["grad_out"][:, q0:q1])).to(device, dt)
    gv, glc, gac, glt, gat = torch.autograd.grad(out, (v_chunk, lc, ac, lt, at), go)
    # the overlapped form (all-gather in flight while the sampling tensors are produced): same results, same gradients
    out2 = cp.sharded_temporal_attention(v_chunk, T, S, shapes, lsi, ftab, lambda: (lc * 1.0, ac * 1.0, lt * 1.0, at * 1.0),
                                         None, None, None)
    g2 = torch.autograd.grad(out2, (v_chunk, lc, ac, lt, at), go)
    # (on the GPU the order of a pixel's terms in grad_value is list order and varies from run to run: last-bit differences)
    same = torch.equal if backend == "gloo" else (lambda a, b: torch.allclose(a, b, rtol=1e-9, atol=1e-12))
    assert same(out2, out) and all(same(a, b) for a, b in zip(g2, (gv, glc, gac, glt, gat))), "overlapped form differs"

    def close(a, b, what):
        err = float(np.abs(a.detach().cpu().numpy() - b).max())
        assert err <= 1e-9 * max(1.0, float(np.abs(b).max())), (what, rank, err)

    close(out, ref[0][:, q0:q1], "out")
    gv_ref = np.zeros_like(flat)
    gv_ref[:rows] = ref[1].reshape(rows, M, D)
    close(gv, gv_ref[rank * chunk:(rank + 1) * chunk], "grad_value chunk")     # summed over ranks
    close(glc, ref[2][:, q0:q1], "grad_loc_c")
    close(gac, ref[3][:, q0:q1], "grad_aw_c")
    close(glt, ref[4][:, q0:q1], "grad_loc_t")
    close(gat, ref[5][:, q0:q1], "grad_aw_t")
    # 16-bit transport of an fp32 clip: value crosses the collective (and is sampled) in bf16, locations / weights stay fp32
    f32 = lambda t: t.detach().to(torch.float32).requires_grad_(True)
    v32, l32 = f32(v_chunk), [f32(t) for t in (lc, ac, lt, at)]
    out3 = cp.sharded_temporal_attention(v32, T, S, shapes, lsi, ftab, *l32, transport_dtype=torch.bfloat16)
    assert out3.dtype == torch.float32
    g3 = torch.autograd.grad(out3, [v32] + l32, go.to(torch.float32))
    assert g3[0].dtype == torch.float32 and all(g.dtype == torch.float32 for g in g3[1:])
    v_b = torch.from_numpy(d["value"]).to(torch.bfloat16).double().numpy()
    ref_b = temporal_reference(v_b, *(d[k] for k in ("shapes", "lsi", "ftab", "loc_c", "aw_c", "loc_t", "aw_t", "grad_out")))
    def near(a, b, tol, what):
        err = float(np.abs(a.detach().double().cpu().numpy() - b).max())
        assert err <= tol * max(1.0, float(np.abs(b).max())), (what, rank, err)
    near(out3, ref_b[0][:, q0:q1], 1e-2, "transport out")             # (the output leaves the op in bf16)
    gvb = np.zeros_like(flat); gvb[:rows] = ref_b[1].reshape(rows, M, D)
    near(g3[0], gvb[rank * chunk:(rank + 1) * chunk], 3e-2, "transport grad_value chunk")    # bf16 partial sums over the ranks
    near(g3[2], ref_b[3][:, q0:q1], 1e-2, "transport grad_aw_c")
    near(g3[4], ref_b[5][:, q0:q1], 1e-2, "transport grad_aw_t")
    # a batch of clips with every all-gather issued up front (clip 1 = clip 0 with other weights): per clip the results of the
    # one-clip calls
    ac2 = (ac.detach() * 0.5).requires_grad_(True)
    outs = cp.sharded_temporal_attention_batch([(v_chunk, lc, ac, lt, at), (v_chunk, lc, ac2, lt, at)], T, S, shapes, lsi, ftab)
    assert same(outs[0], out)
    close(outs[1] - 0.0, ref[0][:, q0:q1] - temporal_reference(d["value"], d["shapes"], d["lsi"], d["ftab"], d["loc_c"], 0.5 * d["aw_c"],
                                                              d["loc_t"], 0.0 * d["aw_t"])[:, q0:q1], "batch clip 1")
    gb = torch.autograd.grad(outs, (v_chunk, lc, ac, lt, at), [go, 0.0 * go])
    assert all(same(a, b) for a, b in zip(gb, (gv, glc, gac, glt, gat))), "batched form differs"
    dist.reduce_scatter_tensor, dist.all_reduce = real_rs, real_ar
    assert calls["rs"] == 5 and calls["ar"] == 0, calls       # plain, overlapped, 16-bit transport, two batched clips
    assert calls["rs_async"] == 3, calls                      # overlapped + two batched clips: issued async, waited where consumed
    # ranks that do not agree on the frame table (or the pyramid) fail -- all of them -- instead of sampling garbage
    if world > 1:
        bad_ftab = ftab.clone()
        if rank == 1:
            bad_ftab[0, 0] = (int(bad_ftab[0, 0]) + 1) % T
        try:
            cp.sharded_temporal_attention(v_chunk, T, S, shapes, lsi, bad_ftab, lc, ac, lt, at)
            raise AssertionError("mismatching frame tables went unnoticed")
        except RuntimeError as e:
            assert "differ between the ranks" in str(e) and "[1]" in str(e), str(e)
        cp.check_ranks_agree(shapes, ftab)                        # the good pair: cached, no collective
    # the ranges tile the query axis
    r = [cp.shard_range(Lq, world, k) for k in range(world)]
    assert r[0][0] == 0 and r[-1][1] == Lq and all(r[i][1] == r[i + 1][0] for i in range(world - 1))
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
