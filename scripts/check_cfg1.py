"""GPU probe: BASELINE configs[1] (encoder 800x1333, N=8, bf16, local sampling) per-kernel times under forced routes."""
import os, sys
os.environ.setdefault("MSDA_ENABLE_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from devis_amd import _native

def run(env, dtype=torch.bfloat16, shapes=bench.PYRAMIDS["B"], N=8, reps=6):
    dev = torch.device("cuda:0")
    S = sum(h * w for h, w in shapes)
    c = bench._plain_op_case(dev, dtype, shapes, N, S, "local", seed=99)
    M, D, P, L, Lq = 8, 32, 4, c["L"], S
    out = torch.empty((N, Lq, M * D), dtype=dtype, device=dev)
    gv = torch.empty(c["value"].shape, dtype=torch.float32, device=dev)
    gl, ga = torch.empty_like(c["loc"]), torch.empty_like(c["aw"])
    ws = _native.bwd_workspace(dev, N, Lq, M, L)
    os.environ.update(env); _native.reload_knobs()
    res = {}
    res["fwd"] = bench._event_ms(lambda: _native.forward(c["value"], c["shapes"], c["lsi"], c["loc"], c["aw"], out), reps)
    routes = [_native.last_route()]
    def bwd():
        ws[:16].zero_()
        rc = _native.load().msda_backward(_native.dtype_code(dtype), c["value"].data_ptr(), c["shapes"].data_ptr(), c["lsi"].data_ptr(),
                                          c["loc"].data_ptr(), c["aw"].data_ptr(), c["grad_out"].data_ptr(), N, S, M, D, L, Lq, P,
                                          gv.data_ptr(), _native.dtype_code(gv.dtype), gl.data_ptr(), ga.data_ptr(), ws.data_ptr(), ws.numel() * 4, None,
                                          _native.shapes_hint(c["shapes"]), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    for ph, key in (("1", "gather"), ("2", "scatter")):
        os.environ["MSDA_BWD_PHASES"] = ph; _native.reload_knobs()
        res[key] = bench._event_ms(bwd, reps)
        routes.append(_native.last_route())
    os.environ.pop("MSDA_BWD_PHASES")
    for k in env: os.environ.pop(k)
    _native.reload_knobs()
    return res, routes, out.float().clone(), gl.float().clone()

if __name__ == "__main__":
    for name, env in (("auto", {}), ("NT=2/tpw=2", {"MSDA_FWD_RS_NT": "2", "MSDA_BWD_RS_TPW": "2"}), ("NT=1/tpw=1", {"MSDA_FWD_RS_NT": "1", "MSDA_BWD_RS_TPW": "1"}),
                      ("NT=4/tpw=4", {"MSDA_FWD_RS_NT": "4", "MSDA_BWD_RS_TPW": "4"})):
        for dt in (torch.bfloat16, torch.float32):
            res, routes, out, gl = run(env, dtype=dt)
            print("%-10s %-8s fwd %.4f gather %.4f scatter %.4f ms" % (name, str(dt).split(".")[1], res["fwd"], res["gather"], res["scatter"]), flush=True)
            print("    ", routes[0], "|", routes[1].split(";")[0])
