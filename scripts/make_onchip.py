"""profiles/onchip.json from the SQ / memory PMC summaries of a profiling round (scripts/profile_round.sh):
    python scripts/make_onchip.py <pmc_sq.txt> <pmc_mem.txt> [tracked name of the sq file] [tracked name of the mem file]
Per kernel of the default bench.py workload, per launch: how busy the units a kernel can be bound by were -- the vector ALU, the LDS
array (and the share of its cycles that were bank conflicts), the matrix pipe, the waves' wait fraction, and the L2 request
count (bench.py turns it into a rate with the launch duration it measures).  Formulas:
    busy cycles of the chip   B = SQ_BUSY_CYCLES / 32 shader engines
    VALU busy                 SQ_INSTS_VALU x 2 cycles / 1024 SIMDs / B
    LDS array busy            SQ_LDS_IDX_ACTIVE / 256 CUs / B          conflicts: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
    matrix pipe busy          SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / B
    wait fraction             SQ_WAIT_ANY / SQ_WAVE_CYCLES
    L2 requests               TCC_REQ_sum (x 128 B per request against the guide's 34.5 TB/s)
The file records the content hash of the kernel sources; bench.py quotes it as `roofline.onchip` only for that very build."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from devis_amd import build


def parse(path):
    cur, vals = None, {}
    for line in open(path):
        m = re.match(r"\S.*?(msda_[a-z_0-9]+kernel)", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"\s+([A-Za-z0-9_]+)\s+\d+\s+per call\s+(\d+)", line)
        if m and cur:
            vals.setdefault(cur, {})[m.group(1)] = int(m.group(2))
    return vals


sq, mem = parse(sys.argv[1]), parse(sys.argv[2])
out = {}
for k, v in sq.items():
    if "zero" in k or "SQ_BUSY_CYCLES" not in v:
        continue
    B = v["SQ_BUSY_CYCLES"] / 32.0
    e = {"busy_cycles": round(B)}
    if "SQ_INSTS_VALU" in v:
        e["valu_busy"] = round(v["SQ_INSTS_VALU"] * 2 / 1024 / B, 4)
    if "SQ_LDS_IDX_ACTIVE" in v:
        e["lds_busy"] = round(v["SQ_LDS_IDX_ACTIVE"] / 256 / B, 4)
        if v["SQ_LDS_IDX_ACTIVE"]:
            e["lds_conflict_share"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"], 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        e["mfma_busy"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / B, 4)
    if "SQ_WAIT_ANY" in v and v.get("SQ_WAVE_CYCLES"):
        e["wait_frac"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4)
    m = mem.get(k, {})
    if "TCC_REQ_sum" in m:
        e["l2_requests"] = m["TCC_REQ_sum"]
        if m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0):
            e["l2_hit"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
    out[k] = e
doc = {
    "_comment": "per launch, default bench.py workload (16 clips, T=6, 300 queries/frame, pyramid A, f32, uniform locations); formulas in "
                "scripts/make_onchip.py; separate rocprofv3 --pmc passes over scripts/step_only.py",
    "source": [sys.argv[3] if len(sys.argv) > 3 else os.path.relpath(os.path.abspath(sys.argv[1]), ROOT),
               sys.argv[4] if len(sys.argv) > 4 else os.path.relpath(os.path.abspath(sys.argv[2]), ROOT)],
    "source_hash": build._source_hash(),
    "workload": {"clips": 16, "frames": 6, "queries": 300, "pyramid": "A", "dtype": "f32", "locs": "uniform", "pattern": "fused"},
    "kernels": out,
}
json.dump(doc, open(os.path.join(ROOT, "profiles", "onchip.json"), "w"), indent=2)
print(json.dumps(out, indent=1))
