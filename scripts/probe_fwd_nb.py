import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import probe_kernels as P
for nb in ("1", "2", "4"):
    os.environ["MSDA_FWD_NB"] = nb
    print("NB", nb)
    for locs in ("uniform", "same"):
        P.run(16, locs)
    P.run(1, "uniform")
