import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import probe_bwd as P
os.environ["MSDA_BWD_PHASES"] = "2"
for cap in ("144", "110", "80", "56"):
    P.run(1, "uniform", env={"MSDA_SCATTER_LDS_KB": cap}, reps=10)
