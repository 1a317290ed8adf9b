"""GPU probe: phase timeline of the owner-computes scatter (a -DMSDA_SCATTER_TRACE build of the library, MSDA_LIB).

    python -m devis_amd.build -DMSDA_SCATTER_TRACE=1 --out=devis_amd/libmsda_exp_trace.so
    MSDA_LIB=$PWD/devis_amd/libmsda_exp_trace.so python scripts/scatter_trace.py [case]

Lane 0 of wave 1 of the first 8 workgroups stamps the shader clock at phase boundaries; the time between two stamps is booked
on the LATER stamp's phase.
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

NAMES = {1: "item start (previous item's tail)", 2: "item decode", 3: "cull batch (records wait, scan, list)", 4: "cull barrier",
         5: "chunks done -> stores start", 6: "stores issued", 7: "item-end barrier", 10: "chunk start", 11: "rows (+ first points) requested",
         12: "taps + links", 13: "rows landed and written", 14: "barrier 1", 15: "point prefetch + walk", 16: "barrier 2"}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "dec16"
    fwd, bwd, gv, reps = scatter_ab.CASES[name]()
    scatter_ab.knobs()
    bwd()
    scatter_ab.knobs(MSDA_BWD_PHASES=2)
    for _ in range(3):
        bwd()
    torch.cuda.synchronize()
    lib = _native.load()
    n = 8192
    buf = (ctypes.c_ulonglong * (8 * n))()
    cnt = (ctypes.c_int * 8)()
    rc = lib.msda_debug_trace(buf, cnt)
    assert rc == n, rc
    arr = np.frombuffer(buf, dtype=np.uint64).reshape(8, n)
    for lo, hi, label in ((0, 4, "wave 1"), (4, 8, "last wave")):
        tot = {}
        span = 0
        chunks = items = 0
        for b in range(lo, hi):
            ev = arr[b, :cnt[b]]
            ids = (ev & np.uint64(255)).astype(np.int64)
            t = (ev >> np.uint64(8)).astype(np.int64)
            dt = np.diff(t)
            for i, d in zip(ids[1:], dt):
                tot[int(i)] = tot.get(int(i), 0) + int(d)
            span += int(t[-1] - t[0]) if len(t) else 0
            chunks += int((ids == 10).sum())
            items += int((ids == 1).sum())
        print("%s, %s: %d items, %d chunks, %.0f clocks per workgroup" % (name, label, items, chunks, span / (hi - lo)))
        for i in sorted(tot):
            print("  %2d %-46s %5.1f %%   %8.0f clk per %s" % (i, NAMES.get(i, "?"), 100.0 * tot[i] / span,
                                                           tot[i] / (chunks if i >= 10 else items), "chunk" if i >= 10 else "item"))


if __name__ == "__main__":
    main()
