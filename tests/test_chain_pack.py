"""CPU: the packed weight stream of the row-chain kernel (lwdetr_amd.kernels.pack_enc_chain) walked lane by lane as
lw-detr_amd/csrc/chain.hip walks it (tests/chain_sim.py) equals the dense formulation of the same chain:
[cv2 + SiLU + LayerNorm ->] memory -> value projections (padding mask on the output rows), enc_output on the rows with invalid
proposals zeroed + LayerNorm, class logits + row maximum (reference: projector.py:117-132, ms_deform_attn.py:110-114,
transformer.py:113-116, :231-246)."""
import numpy as np
import pytest
import torch

import lwdetr_amd  # noqa: F401
from lwdetr_amd import kernels as K
from chain_sim import simulate_enc_wave


def _ln(x, g, b, eps):
    mu = x.mean(1, keepdims=True)
    var = ((x - mu) ** 2).mean(1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


@pytest.mark.parametrize("d,k5,nl", [(256, 640, 3), (256, 0, 3), (384, 0, 2)])
def test_enc_chain_stream_walk_equals_dense(d, k5, nl):
    g = torch.Generator().manual_seed(d + k5)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float32).double()      # f32-representable: the packer keeps f32 masters
    ncls = 91
    w_enc, b_enc, g_enc, be_enc = r(d, d) / 16, r(d), (1 + 0.125 * r(d)).float().double(), 0.125 * r(d)
    w_cls, b_cls = r(ncls, d) / 16, r(ncls)
    w_val, b_val = r(nl * d, d) / 16, r(nl * d)
    cv2 = ((r(d, k5) / 32).float().double(), r(d), (1 + 0.125 * r(d)).float().double(), 0.125 * r(d)) if k5 else None
    stream, vec = K.pack_enc_chain(d, torch.float64, w_enc, b_enc, g_enc, be_enc, w_cls, b_cls, w_val, b_val, cv2=cv2)
    # sizes as the C ABI helpers compute them (restated: no library needed on the CPU)
    pieces = ((d // 32) * (k5 // 64) if k5 else 0) + (nl * (d // 32) + d // 32 + 3) * (d // 64) + 2
    assert stream.numel() == pieces * 2048
    assert vec.numel() % 1024 == 0 and vec.numel() >= (3 * d if k5 else 0) + 3 * d + 96 + 6 * d
    x = r(32, k5 or d).numpy()
    rowvalid = (torch.rand(32, generator=g) > 0.3).numpy().astype(np.float64)
    notpad = (torch.rand(32, generator=g) > 0.2).numpy().astype(np.float64)
    eps_p, eps_e = 1e-6, 1e-5
    sim = simulate_enc_wave(stream.double().numpy(), vec.double().numpy(), x, rowvalid, notpad, d, k5, nl, ncls, eps_p, eps_e)
    assert sim["fragments_consumed"] == (pieces - 2) * 4
    n = lambda t: t.double().numpy()
    if k5:
        z = x @ n(cv2[0]).T + n(cv2[1])
        mem = _ln(z / (1 + np.exp(-z)), n(cv2[2]), n(cv2[3]), eps_p)
        np.testing.assert_allclose(sim["memory"], mem, rtol=0, atol=1e-9)
    else:
        mem = x
    vals = (mem @ n(w_val).T + n(b_val)) * notpad[:, None]
    np.testing.assert_allclose(sim["values"], vals.reshape(32, nl, d).transpose(1, 0, 2), rtol=0, atol=1e-9)
    om = _ln((mem * rowvalid[:, None]) @ n(w_enc).T + n(b_enc), n(g_enc), n(be_enc), eps_e)
    np.testing.assert_allclose(sim["om"], om, rtol=0, atol=1e-9)
    cls = om @ n(w_cls).T + n(b_cls)
    np.testing.assert_allclose(sim["cls"][:, :ncls], cls, rtol=0, atol=1e-9)
    assert np.abs(sim["cls"][:, ncls:]).max() == 0
    np.testing.assert_allclose(sim["cls_max"], cls.max(1), rtol=0, atol=1e-9)


@pytest.mark.parametrize("d", [256, 384])
def test_row_chain_split_stream_walk_equals_dense(d):
    """RowChainOp.pack walked as mlp_chain_split_kernel walks it (tiles dealt over 4 waves; d = 384: two k-half steps per tile, k-half-major
    stream) equals the dense chain: class head from the input, out_proj + residual + LayerNorm (+ query_pos), a 100-column side stage, an MLP stage."""
    from chain_sim import simulate_row_chain_split
    g = torch.Generator().manual_seed(d)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float32).double()
    n = lambda t: t.double().numpy()
    w0, b0 = r(91, d) / 16, r(91)
    w1, b1, g1, be1 = r(d, d) / 16, r(d), (1 + 0.125 * r(d)).float().double(), 0.125 * r(d)
    w2, b2 = r(100, d) / 16, r(100)
    w3, b3 = r(d, d) / 16, r(d)
    stages = [dict(kind="side", w=w0, b=b0), dict(kind="full", w=w1, b=b1, res=True, ln=(g1, be1, 1e-5), addq=True),
              dict(kind="side", w=w2, b=b2), dict(kind="full", w=w3, b=b3, relu=True)]
    stream, vec = K.RowChainOp.pack(d, torch.float64, d, stages)
    x, res, q = n(r(32, d)), n(r(32, d)), n(r(32, d))
    sim_stages = [dict(kind="side", n=91), dict(kind="full", res=True, ln=1e-5, addq=True), dict(kind="side", n=100), dict(kind="full", relu=True)]
    outs, pieces = simulate_row_chain_split(stream.double().numpy(), vec.double().numpy(), sim_stages, x, res, q, d)
    assert pieces * 2048 + 2 * 2048 == stream.numel()
    y0 = x @ n(w0).T + n(b0)
    y1 = _ln(x @ n(w1).T + n(b1) + res, n(g1), n(be1), 1e-5)
    y2 = (y1 + q) @ n(w2).T + n(b2)
    y3 = np.maximum((y1 + q) @ n(w3).T + n(b3), 0.0)
    for got, ref in zip(outs, (y0, y1, y2, y3)):
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9)
