import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import probe_kernels as P
for dbg in ("0", "32"):
    os.environ["MSDA_DBG"] = dbg
    print("MSDA_DBG", dbg); P.run(16, "uniform"); P.run(16, "clustered"); P.run(1, "uniform")
