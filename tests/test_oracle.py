"""Pins the CPU oracle (oracle/) to the golden vectors made from the reference's own Python oracle
``ms_deform_attn_core_pytorch`` + autograd (tests/golden/make_golden.py).  CPU only.

The reference ships no numeric known-answer vectors for this path (src/models/ops/test.py only
prints allclose/gradcheck verdicts), so these fixtures -- outputs of the reference itself, generated
in the build container -- are the pin."""
import numpy as np
import pytest
import torch

from conftest import OP_FIXTURES, golden


@pytest.mark.parametrize("name", OP_FIXTURES)
def test_c_oracle_forward_fp64_matches_reference(oracle, name):
    g = golden(name)
    out = oracle.forward(g["value"].astype(np.float64), g["spatial_shapes"], g["level_start_index"],
                         g["sampling_locations"].astype(np.float64),
                         g["attention_weights"].astype(np.float64))
    # reference test.py:40 uses torch.allclose defaults (rtol 1e-5, atol 1e-8) in fp64; we are far tighter
    np.testing.assert_allclose(out, g["out"], rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("name", OP_FIXTURES)
def test_c_oracle_backward_fp64_matches_reference_autograd(oracle, name):
    g = golden(name)
    gv, gl, ga = oracle.backward(g["value"].astype(np.float64), g["spatial_shapes"],
                                 g["level_start_index"], g["sampling_locations"].astype(np.float64),
                                 g["attention_weights"].astype(np.float64), g["grad_output"])
    np.testing.assert_allclose(gv, g["grad_value"], rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(ga, g["grad_attn_weight"], rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(gl, g["grad_sampling_loc"], rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("name", OP_FIXTURES)
def test_c_oracle_forward_fp32(oracle, name):
    g = golden(name)
    out = oracle.forward(g["value"], g["spatial_shapes"], g["level_start_index"],
                         g["sampling_locations"], g["attention_weights"])
    assert out.dtype == np.float32
    # BASELINE.json: <= 1e-4 max abs in fp32 (reference test.py:56 only asks rtol 1e-2 / atol 1e-3)
    assert np.abs(out - g["out"]).max() <= 1e-6
    assert np.abs(out - g["out_f32"]).max() <= 1e-6


@pytest.mark.parametrize("name", OP_FIXTURES)
def test_torch_restatement_matches_reference(oracle, name):
    """oracle.grid_sample_forward is what bench.py times as the CPU baseline."""
    g = golden(name)
    v, l, a = (torch.from_numpy(g[k].astype(np.float64)).requires_grad_(True)
               for k in ("value", "sampling_locations", "attention_weights"))
    out = oracle.grid_sample_forward(v, torch.from_numpy(g["spatial_shapes"]), l, a)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-12, atol=1e-15)
    gv, gl, ga = torch.autograd.grad(out, (v, l, a), torch.from_numpy(g["grad_output"]))
    np.testing.assert_allclose(gv.numpy(), g["grad_value"], rtol=1e-11, atol=1e-14)
    # exactly on h_im == -1 / w_im == -1 grid_sample differentiates one-sidedly while the reference
    # CUDA kernel skips the point (cuh:288); the golden follows the kernel (see make_golden.py)
    keep = ~g["on_minus_one_edge"][..., None]
    np.testing.assert_allclose(gl.numpy() * keep, g["grad_sampling_loc"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(ga.numpy(), g["grad_attn_weight"], rtol=1e-11, atol=1e-14)


def test_reference_testpy_procedure(oracle):
    """The reference's own check (src/models/ops/test.py:30-58), restated with asserts: forward fp64
    allclose at torch defaults, forward fp32 at rtol 1e-2 / atol 1e-3, on its shapes and seed."""
    g = golden("op_testpy_shape")
    assert g["value"].shape == (1, 30, 2, 2) and g["sampling_locations"].shape == (1, 2, 2, 2, 2, 2)
    o64 = oracle.forward(g["value"].astype(np.float64), g["spatial_shapes"], g["level_start_index"],
                         g["sampling_locations"].astype(np.float64),
                         g["attention_weights"].astype(np.float64))
    assert np.allclose(o64, g["out"], rtol=1e-5, atol=1e-8)
    o32 = oracle.forward(g["value"], g["spatial_shapes"], g["level_start_index"],
                         g["sampling_locations"], g["attention_weights"])
    assert np.allclose(o32, g["out_f32"], rtol=1e-2, atol=1e-3)


def test_skipped_points_have_zero_gradients(oracle):
    """cuh:288 + zero-filled grads (ms_deform_attn_cuda.cu:121-123)."""
    g = golden("op_out_of_range")
    H, W = g["spatial_shapes"][0]
    loc = g["sampling_locations"].astype(np.float64)
    y = loc[0, :, 0, 0, 0, 1] * H - 0.5
    outside = ~((y > -1) & (y < H))
    assert outside.sum() >= 4            # the fixture really contains skipped points
    assert np.all(g["grad_attn_weight"][0, outside, 0, 0, 0] == 0)
    assert np.all(g["grad_sampling_loc"][0, outside, 0, 0, 0] == 0)
