import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import probe_bwd as P
os.environ["MSDA_BWD_PHASES"] = "2"
for dbg in ("0","2","4"):
    P.run(16, "uniform", env={"MSDA_SCATTER_DBG": dbg}, reps=5)
P.run(16, "clustered"); P.run(1, "uniform"); P.run(8, "uniform", pyramid="B")
