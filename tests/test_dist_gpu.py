"""GPU: the RCCL (backend "nccl") branch of Mode 2 with the real HIP kernels, world_size 1 (the GPU box
has one GPU; collectives degenerate but the reduce_scatter_tensor / all_gather_into_tensor code path runs)."""
import pytest

from test_dist_cpu import _launch

pytestmark = pytest.mark.gpu


def test_sharded_clip_rccl_single_rank():
    _launch(1, "nccl", timeout=600)
