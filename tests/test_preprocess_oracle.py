"""CPU: the preprocessing oracle (oracle/preprocess_ref.py) pinned bit-exactly against Pillow's output (the library behind the
reference's SquareResize), and the product's host-side coefficient tables against the oracle's literal loops."""
import os

import numpy as np
import pytest
import torch

from helpers import ROOT
from oracle import preprocess_ref as P

GOLD = os.path.join(ROOT, "tests", "golden", "preprocess.npz")


def _cases():
    g = np.load(GOLD)
    n = sum(1 for k in g.files if k.startswith("in_"))
    return g, n


def test_oracle_resize_matches_pillow_golden_bit_exact():
    g, n = _cases()
    assert n >= 8
    for i in range(n):
        out = P.square_resize_u8(g[f"in_{i}"], int(g[f"size_{i}"]))
        assert out.dtype == np.uint8 and np.array_equal(out, g[f"out_{i}"]), i


def test_oracle_resize_matches_live_pillow_when_importable():
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    rng = np.random.default_rng(7)
    for (h, w, s) in [(480, 640, 640), (427, 640, 640), (333, 500, 128), (720, 1280, 320), (61, 47, 64)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((s, s), Image.BILINEAR))
        assert np.array_equal(P.square_resize_u8(img, s), ref), (h, w, s, PIL.__version__)


def test_oracle_to_tensor_normalize_is_the_reference_arithmetic():
    u8 = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2)
    x = P.to_tensor_normalize(u8)
    ref = torch.from_numpy(u8).permute(2, 0, 1).float().div(255)
    for c, (m, s) in enumerate(zip(P.MEAN, P.STD)):
        assert torch.equal(x[c], (ref[c] - m) / s)          # F.normalize: tensor.sub_(mean).div_(std), f32
    assert P.to_tensor_normalize(u8, torch.float16).dtype == torch.float16


@pytest.mark.parametrize("in_out", [(640, 640), (480, 640), (427, 640), (1280, 640), (500, 128), (20, 64), (64, 640), (9, 32),
                                     (2000, 64), (333, 333)])
def test_host_tables_equal_oracle_loops(in_out):
    import lwdetr_amd.preprocess as pp
    n_in, n_out = in_out
    b0, k0 = P.resample_coeffs(n_in, n_out)
    b1, k1 = pp.resample_tables(n_in, n_out)
    assert np.array_equal(b0, b1) and np.array_equal(k0, k1)
    assert k1.dtype == np.int32 and (k1.sum(1) - (1 << P.PRECISION_BITS)).__abs__().max() <= k1.shape[1]    # taps sum to ~1.0
