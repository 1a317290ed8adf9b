"""GPU: the RCCL (backend "nccl") branch of Mode 2 with the real HIP kernels, world_size 1 (the GPU box
has one GPU; collectives degenerate but the reduce_scatter_tensor / all_gather_into_tensor code path runs)."""
import os

import pytest

from conftest import ROOT
from test_dist_cpu import _launch

pytestmark = pytest.mark.gpu


def test_sharded_clip_rccl_single_rank():
    _launch(1, "nccl", timeout=600)


def test_bench_n_rank_protocol_on_one_gpu():
    """`python bench.py --gpus 2` end to end on the GPU box: the launcher starts two ranks (child torchrun), they rendezvous, run the
    warm-up and timed steps between barriers, reduce the time with MAX and count themselves -- everything an 8-GPU run does except
    that both ranks share cuda:0 and talk over gloo (`--share-device`: RCCL refuses two ranks on one device).  The line must say
    n_gpus 2, ranks_seen 2, carry the test marker, and its whole-job rate must be the two ranks' rows over the slower rank's time."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--share-device", "--clips", "4",
                        "--steps", "5", "--warmup", "2", "--no-other-configs", "--no-cpu-baseline", "--prewarm-seconds", "0.1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and "test_run" in line and line["scaling"] == "weak"
    rows = 2 * 4 * 6 * 300
    assert line["config"]["query_rows_per_step"] == rows
    assert abs(line["value"] - rows / (line["ms_per_step"] * 1e-3) / 1e6) <= 0.01 * line["value"]
    assert line["roofline"] is not None and line["roofline"]["frac"] > 0          # rank 0's per-kernel timings still there
