"""profiles/hbm_traffic.json from the HBM PMC summary of a profiling round (scripts/profile_round.sh):
    python scripts/make_hbm_traffic.py profiles/r03_x_pmc_hbm_traffic.txt [path to record as the source]
(the second argument: on the GPU box the summary still sits in the scratch directory gpurun_out/; the file records the
tracked name it is committed under by scripts/collect_round.sh, so that `roofline.traffic_source` resolves in the repository)
FETCH_SIZE / WRITE_SIZE are in KB per launch (separate rocprofv3 --pmc passes over scripts/step_only.py, the default
bench.py workload).  The file records the content hash of the kernel sources it was measured on; bench.py quotes it as
`roofline.traffic` only while that hash is the hash of the build it runs."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from devis_amd import build

src = sys.argv[1]
cur, vals = None, {}
for line in open(src):
    m = re.match(r"\S.*?(msda_[a-z_0-9]+kernel)", line)
    if m:
        cur = m.group(1)
        continue
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+\d+\s+per call\s+(\d+)", line)
    if m and cur:
        vals.setdefault(cur, {})[m.group(1)] = int(m.group(2)) * 1024
out = {
    "_comment": "HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs over "
                "scripts/step_only.py, KB units x 1024) for the default bench.py workload: 16 clips, T=6, 300 queries/frame, pyramid A, "
                "f32, uniform locations, fused pattern. FETCH_SIZE is reported raw (= TCC_EA0_RDREQ x 64 B; MI355X_MICROARCH.md: wide "
                "streaming reads are 128-byte requests tallied at 64, so the true read volume lies between 1x and 2x this figure). "
                "WRITE_SIZE calibrates exactly on the known output sizes.",
    "source": sys.argv[2] if len(sys.argv) > 2 else os.path.relpath(os.path.abspath(src), ROOT),
    "source_hash": build._source_hash(),
    "workload": {"clips": 16, "frames": 6, "queries": 300, "pyramid": "A", "dtype": "f32", "locs": "uniform", "pattern": "fused"},
    "kernels": {k: {"fetch_bytes": v.get("FETCH_SIZE", 0), "write_bytes": v.get("WRITE_SIZE", 0)} for k, v in vals.items() if "zero" not in k},
}
json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=2)
print(json.dumps(out["kernels"], indent=1))
