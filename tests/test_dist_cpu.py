"""Multi-process CPU tests (gloo, world_size 2 and 3) of the N>1 paths: Mode 2 sharding
(devis_amd/clip_parallel.py) and bench.py's clip-parallel bookkeeping."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(world, backend, timeout=240, case="small"):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), backend, case]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for k in range(world):
        assert "rank %d ok" % k in r.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_clip_gloo(world):
    _launch(world, "gloo")


def test_sharded_clip_gloo_world8_decoder_clip_shape():
    """BASELINE configs[3] as far as it runs without the 8-GPU node: 8 ranks, the T = 6 decoder clip on the 360x640
    pyramid (frames do not divide the ranks: the cut is on pixel rows and queries), connect-all window, M = 8 x D = 32;
    plain and overlapped all-gather; every rank's outputs and gradients against the oracle."""
    _launch(8, "gloo", timeout=600, case="cfg3")


def test_shard_range_properties():
    from devis_amd.clip_parallel import padded_chunk, shard_range
    for n in (0, 1, 7, 300, 1800):
        for world in (1, 2, 3, 8):
            rs = [shard_range(n, world, r) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n
            assert padded_chunk(n, world) * world >= n
