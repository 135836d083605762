"""CPU: the C restatement of the reference CUDA forward kernel reproduces the reference's own KAT vectors."""
import numpy as np
import torch

from oracle import msda_c
from oracle import lwdetr_torch as O
from helpers import load_golden


def test_c_oracle_matches_reference_kat_double():
    g = load_golden("msda_op_kat")
    out = msda_c.msda_forward(g["double_value"], g["shapes"], g["double_loc"], g["double_aw"])
    # models/ops/test.py:56 uses torch.allclose defaults (rtol 1e-5, atol 1e-8) in double
    np.testing.assert_allclose(out, g["double_out"], rtol=1e-5, atol=1e-8)
    assert np.allclose(np.round(out, 4), [[[0.0019, 0.0046, 0.0047, 0.0044], [0.0038, 0.0025, 0.0018, 0.0036]]])


def test_c_oracle_matches_reference_kat_float():
    g = load_golden("msda_op_kat")
    out = msda_c.msda_forward(g["float_value"], g["shapes"], g["float_loc"], g["float_aw"])
    np.testing.assert_allclose(out, g["float_out"], rtol=1e-2, atol=1e-3)     # models/ops/test.py:82
    np.testing.assert_allclose(out, g["float_out"], rtol=0, atol=1e-7)


def test_c_oracle_out_of_bounds_branches():
    g = load_golden("msda_op_kat")
    out = msda_c.msda_forward(g["oob_value"], g["oob_shapes"], g["oob_loc"], g["oob_aw"])
    np.testing.assert_allclose(out, g["oob_out"], rtol=0, atol=2e-6)
    # the torch restatement's core agrees too (third independent implementation)
    shapes = [tuple(r) for r in g["oob_shapes"].tolist()]
    out_t = O.msda_core(torch.from_numpy(g["oob_value"]), shapes, torch.from_numpy(g["oob_loc"]),
                        torch.from_numpy(g["oob_aw"]))
    np.testing.assert_allclose(out_t.numpy(), g["oob_out"], rtol=0, atol=2e-6)


def test_c_oracle_empty_and_ragged():
    shapes = np.array([[3, 5]], dtype=np.int64)
    value = np.random.default_rng(0).standard_normal((1, 15, 1, 4)).astype(np.float32)
    loc = np.zeros((1, 0, 1, 1, 2, 2), np.float32)
    aw = np.zeros((1, 0, 1, 1, 2), np.float32)
    assert msda_c.msda_forward(value, shapes, loc, aw).shape == (1, 0, 4)
    # a sample exactly on a pixel centre returns that pixel; far outside returns 0
    loc = np.array([[(2 + 0.5) / 5, (1 + 0.5) / 3], [5.0, 5.0]], np.float32).reshape(1, 1, 1, 1, 2, 2)
    aw = np.array([1.0, 1.0], np.float32).reshape(1, 1, 1, 1, 2)
    out = msda_c.msda_forward(value, shapes, loc, aw)
    np.testing.assert_allclose(out[0, 0], value[0, 1 * 5 + 2, 0], atol=1e-6)


def test_c_oracle_backward_matches_reference_autograd():
    """col2im restatement vs autograd through the reference's PyTorch core (tests/golden/msda_op_grad_kat.npz)."""
    g = load_golden("msda_op_grad_kat")
    for tag in ("kat", "oob", "c30", "c71"):
        gv, gl, ga = msda_c.msda_backward(g[f"{tag}_value"], g[f"{tag}_shapes"], g[f"{tag}_loc"], g[f"{tag}_aw"],
                                          g[f"{tag}_grad_out"])
        np.testing.assert_allclose(gv, g[f"{tag}_grad_value"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(ga, g[f"{tag}_grad_aw"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(gl, g[f"{tag}_grad_loc"], rtol=1e-8, atol=1e-11)
        # float instantiation within float round-off of the double gradients
        gv32, gl32, ga32 = msda_c.msda_backward(*(g[f"{tag}_{k}"].astype(np.float32) for k in ("value",)), g[f"{tag}_shapes"],
                                                *(g[f"{tag}_{k}"].astype(np.float32) for k in ("loc", "aw", "grad_out")))
        scale = max(1.0, float(np.abs(g[f"{tag}_grad_loc"]).max()))
        assert np.abs(gv32 - g[f"{tag}_grad_value"]).max() < 1e-5
        assert np.abs(ga32 - g[f"{tag}_grad_aw"]).max() < 1e-4 * max(1.0, float(np.abs(g[f"{tag}_grad_aw"]).max()))
        assert np.abs(gl32 - g[f"{tag}_grad_loc"]).max() < 2e-4 * scale
