"""CPU: host packing of the fused ViT block kernel (lwdetr_amd.kernels.pack_vit_block) against a lane-level emulation of
lw-detr_amd/csrc/vitblock.hip (tests/vitblock_sim.py) and the dense formulation of models/backbone/vit.py:195-222."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.vitblock_sim import gelu_fast16, simulate_wave


def _weights(c, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g, dtype=torch.float64) * sc
    return dict(wp=r(c, c, sc=c ** -0.5), bp=r(c, sc=0.1), g1=r(c, sc=0.3) + 0.5, w1=r(4 * c, c, sc=c ** -0.5), b1=r(4 * c, sc=0.1),
                w2=r(c, 4 * c, sc=(4 * c) ** -0.5), b2=r(c, sc=0.1), g2=r(c, sc=0.3) - 0.6, ln2_w=r(c, sc=0.2) + 1, ln2_b=r(c, sc=0.1),
                wqkv=r(3 * c, c, sc=c ** -0.5), qb=r(c, sc=0.1), vb=r(c, sc=0.1), ln1_w=r(c, sc=0.2) + 1, ln1_b=r(c, sc=0.1))


def _dense(w, x, att, c, eps=1e-6):
    x1 = x + w["g1"] * (att @ w["wp"].t() + w["bp"])
    hid = F.layer_norm(x1, (c,), w["ln2_w"], w["ln2_b"], eps) @ w["w1"].t() + w["b1"]
    hid = torch.from_numpy(gelu_fast16(hid.numpy()))
    xn = x1 + w["g2"] * (hid @ w["w2"].t() + w["b2"])
    y = F.layer_norm(xn, (c,), w["ln1_w"], w["ln1_b"], eps) @ w["wqkv"].t() + torch.cat([w["qb"], torch.zeros_like(w["qb"]), w["vb"]])
    return xn, y


@pytest.mark.parametrize("c,nh,t0,nvalid,order", [(192, 1, 8, 32, "alternating"), (192, 1, 48, 24, "alternating"), (192, 2, 48, 64, "pipelined"),
                                                    (384, 1, 24, 24, "pipelined")])
def test_pack_vit_block_matches_dense_through_lane_emulation(c, nh, t0, nvalid, order):
    from lwdetr_amd import kernels as K
    w = _weights(c)
    m, tp, heads = 120, 60, c // 32
    hd = c // heads
    g = torch.Generator().manual_seed(5)
    x = torch.randn(m, c, generator=g, dtype=torch.float64) * 1.5 + 0.2
    att = torch.randn(m, c, generator=g, dtype=torch.float64)
    stream, vec = K.pack_vit_block(w["wp"], w["bp"], w["g1"], w["w1"], w["b1"], w["w2"], w["b2"], w["g2"], w["ln2_w"], w["ln2_b"],
                                   torch.float64, qkv=(w["wqkv"], w["qb"], w["vb"], w["ln1_w"], w["ln1_b"]), order=order)
    nti = c // 32
    assert stream.numel() == (nti + 8 * nti + 3 * nti + (2 if order == "alternating" else 0)) * (c // 16) * 512
    out, writes = simulate_wave(stream.numpy(), vec.numpy(), x.numpy(), att.numpy(), t0, nvalid, c, nh, 1e-6, 1e-6,
                                qkv=dict(heads=heads, hd=hd, Tp=tp, qscale=0.37), order=order)
    xn, y = _dense(w, x, att, c)
    assert np.abs(out - xn[t0:t0 + nvalid].numpy()).max() < 5e-6       # the packer keeps f32 master copies
    # every q / k / v^T element of the wave's tokens is written exactly once, at the address of the HEADS / HEADS_T layouts
    nb = m // tp
    sp = lambda t_: t_.reshape(nb, tp, heads, hd).permute(0, 2, 1, 3).contiguous()
    q_ref, k_ref = (sp(y[:, :c]) * 0.37).reshape(-1), sp(y[:, c:2 * c]).reshape(-1)
    v_ref = sp(y[:, 2 * c:]).transpose(2, 3).contiguous().reshape(-1)
    assert len(writes) == 3 * nvalid * c
    for (kind, idx), val in writes.items():
        ref = {"q": q_ref, "k": k_ref, "v": v_ref}[kind][idx].item()
        assert abs(val - ref) < 5e-6, (kind, idx, val, ref)
    # ... and those addresses are the ones of tokens t0 .. t0 + nvalid - 1
    tok_q = {idx // hd % tp + (idx // (hd * tp * heads)) * tp for (kind, idx) in writes if kind == "q"}
    assert tok_q == set(range(t0, t0 + nvalid))


def test_pack_vit_block_refuses_zero_layerscale():
    from lwdetr_amd import kernels as K
    w = _weights(192)
    w["g2"][7] = 0.0
    with pytest.raises(ValueError):
        K.pack_vit_block(w["wp"], w["bp"], w["g1"], w["w1"], w["b1"], w["w2"], w["b2"], w["g2"], w["ln2_w"], w["ln2_b"], torch.float16)


def test_packed_f16_gelu_model_error_budget():
    """The opt-in packed-f16 GELU of the block kernel (vitblock_sim.gelu_vb16_packed = the instruction sequence with every result rounded
    to f16) against the exact erf form over every f16 input in [-12, 12] and on N(0, 1.5) inputs: saturates correctly at both ends (no NaN
    from the f16 overflow of x^2 or 2^t), maximum error 2.7e-3 (1.4 f16 ulps of a result near 3), rms 4.2e-4 - 1.55x the f32-arithmetic form
    (2.7e-4), of which 2.0e-4 is the f16 rounding of the result that both share."""
    from scipy.special import erf
    from tests.vitblock_sim import gelu_vb16, gelu_vb16_packed
    exact = lambda v: 0.5 * v * (1.0 + erf(v / np.sqrt(2.0)))
    allh = np.arange(65536, dtype=np.uint32).astype(np.uint16).view(np.float16)
    allh = allh[np.isfinite(allh)].astype(np.float64)
    y = gelu_vb16_packed(allh)
    assert np.isfinite(y).all()
    big = np.abs(allh) > 12
    assert np.array_equal(y[big & (allh > 0)], allh[big & (allh > 0)]) and np.all(y[big & (allh < 0)] == 0)
    x = allh[~big]
    assert np.abs(gelu_vb16_packed(x) - exact(x)).max() < 2.8e-3
    xs = (np.random.default_rng(0).standard_normal(200000) * 1.5).astype(np.float16).astype(np.float64)
    rms = lambda e: float(np.sqrt((e ** 2).mean()))
    r_packed = rms(gelu_vb16_packed(xs) - exact(xs))
    r_f32 = rms(gelu_vb16(xs).astype(np.float16).astype(np.float64) - exact(xs))
    assert r_packed < 4.5e-4 and r_packed < 1.7 * r_f32, (r_packed, r_f32)

