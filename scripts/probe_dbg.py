import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import probe_bwd as P
for dbg in ("0","1","2","4"):
    P.run(16, "uniform", env={"MSDA_SCATTER_DBG": dbg}, reps=3)
P.run(1, "uniform"); P.run(4, "uniform"); P.run(16, "clustered"); P.run(16, "uniform", dtype="bf16"); P.run(8, "uniform", pyramid="B")
P.run(16, "uniform", env={"MSDA_SCATTER_LDS_KB": "72", "MSDA_SCATTER_WG_PER_CU": "2"})
