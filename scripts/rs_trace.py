"""GPU probe: phase timeline of the resident-slab forward (a -DMSDA_RS_TRACE build of the library, MSDA_LIB).

    python -m devis_amd.build -DMSDA_RS_TRACE=1 --out=devis_amd/libmsda_exp_rstrace.so
    MSDA_LIB=$PWD/devis_amd/libmsda_exp_rstrace.so python scripts/rs_trace.py [case]
"""
import ctypes
import os
import sys

os.environ.setdefault("MSDA_ENABLE_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import scatter_ab
from devis_amd import _native

NAMES = {9: "prologue", 1: "barrier: previous slab free", 2: "slab pieces issued + landed (vmcnt 0)", 3: "barrier: slab complete",
         4: "slot: point loads issued + transposed", 5: "slot: corners (records, LDS + level-0 gathers)", 6: "last tile -> end of frames"}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "dec16"
    fwd, bwd, gv, reps = scatter_ab.CASES[name]()
    scatter_ab.knobs()
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    lib = _native.load()
    n = 4096
    buf = (ctypes.c_ulonglong * (8 * n))()
    cnt = (ctypes.c_int * 8)()
    rc = lib.msda_debug_trace_rs(buf, cnt)
    assert rc == n, rc
    arr = np.frombuffer(buf, dtype=np.uint64).reshape(8, n)
    for lo, hi, label in ((0, 4, "wave 1"), (4, 8, "last wave")):
        tot, num = {}, {}
        span = 0
        for b in range(lo, hi):
            ev = arr[b, :cnt[b]]
            ids = (ev & np.uint64(255)).astype(np.int64)
            t = (ev >> np.uint64(8)).astype(np.int64)
            for i, d in zip(ids[1:], np.diff(t)):
                tot[int(i)] = tot.get(int(i), 0) + int(d)
                num[int(i)] = num.get(int(i), 0) + 1
            span += int(t[-1] - t[0]) if len(t) else 0
        print("%s, %s: %.0f clocks per workgroup" % (name, label, span / (hi - lo)))
        for i in sorted(tot):
            print("  %2d %-52s %5.1f %%   %8.0f clk x %d" % (i, NAMES.get(i, "?"), 100.0 * tot[i] / span, tot[i] / num[i], num[i] // (hi - lo)))


if __name__ == "__main__":
    main()
