"""GPU: every HIP kernel of liblwdetr_hip.so against a plain PyTorch fp32 reference of the same op."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.float16, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.float16: 4e-3, torch.bfloat16: 3e-2}


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    return torch.device("cuda:0")


def _rand(*shape, dtype=torch.float32, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(_dev())


def _need_experiments():
    """The measured-slower opt-in kernels (4-wave large-tile form, LayerNorm-folded epilogue, split-K, LDS window tiles) are compiled only with
    `make TUNE=-DLWDETR_EXPERIMENTS` since round 6; their tests run against such a build (LWDETR_HIP_LIB) and skip on the default library."""
    from lwdetr_amd import _native
    if not _native.lib().lwdetr_has_experiments():
        pytest.skip("kernel variant not in the default build (make TUNE=-DLWDETR_EXPERIMENTS)")


def _relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mnk", [(300, 192, 192), (1000, 91, 256), (257, 4, 256), (4096, 576, 192), (513, 768, 2048),
                                 # BASELINE row counts (32 images x 1600 tokens and a ragged one): the A-panel-resident schedule
                                 (51200, 256, 768), (51200, 91, 256), (33001, 576, 192), (40000, 256, 640)])
def test_gemm_linear_bias_act_res(dtype, mnk):
    from lwdetr_amd import kernels as K
    m, n, k = mnk
    x, w = _rand(m, k, dtype=dtype, seed=1), _rand(n, k, dtype=dtype, scale=k ** -0.5, seed=2)
    b, g = _rand(n, seed=3), _rand(n, seed=4)
    r = _rand(m, n, dtype=dtype, seed=5)
    for act, fn in [(K.ACT_NONE, lambda t: t), (K.ACT_RELU, F.relu), (K.ACT_GELU, F.gelu), (K.ACT_SILU, F.silu)]:
        out = K.linear(x, w, b, act=act)
        ref = fn(x.float() @ w.float().t() + b)
        assert _relerr(out, ref) < TOL[dtype], (act, _relerr(out, ref))
    out = K.linear(x, w, b, res=r, gamma=g)
    ref = r.float() + g * (x.float() @ w.float().t() + b)
    assert _relerr(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hd,b,tp", [(16, 2, 104), (32, 2, 104), (64, 2, 104), (16, 32, 1600), (32, 24, 1600)])
def test_gemm_qkv_head_layouts(dtype, hd, b, tp):
    """Three column segments: Q (HEADS, bias, scale), K (HEADS), V^T (HEADS_T, bias) - transpose-detecting data.
    The (32, 1600) / (24, 1600) cases are full-size batches (block 0 of small / medium: A-panel-resident schedule)."""
    _check_qkv_layouts(dtype, hd, b, tp)


def _check_qkv_layouts(dtype, hd, b, tp):
    from lwdetr_amd import kernels as K
    heads = 12
    c = heads * hd
    x = _rand(b * tp, c, dtype=dtype, seed=1)
    w = _rand(3 * c, c, dtype=dtype, scale=c ** -0.5, seed=2)
    qb, vb = _rand(c, seed=3), _rand(c, seed=4)
    q = torch.zeros(b, heads, tp, hd, dtype=dtype, device=_dev())
    k = torch.zeros_like(q)
    vt = torch.zeros(b, heads, hd, tp, dtype=dtype, device=_dev())
    K.GemmOp(x, w, b * tp, 3 * c, c, [
        K.seg(q, 0, c, mode=K.OUT_HEADS, bias=qb, scale=0.37, p0=tp, p1=hd, p2=heads),
        K.seg(k, c, 2 * c, mode=K.OUT_HEADS, p0=tp, p1=hd, p2=heads),
        K.seg(vt, 2 * c, 3 * c, mode=K.OUT_HEADS_T, bias=vb, p0=tp, p1=hd, p2=heads)])()
    y = x.float() @ w.float().t()
    sp = lambda t: t.reshape(b, tp, heads, hd).permute(0, 2, 1, 3)
    assert _relerr(q, sp((y[:, :c] + qb) * 0.37)) < TOL[dtype]
    assert _relerr(k, sp(y[:, c:2 * c])) < TOL[dtype]
    assert _relerr(vt, sp(y[:, 2 * c:] + vb).transpose(2, 3)) < TOL[dtype]


def _to_winmajor(x_bhwc, twp):
    """(B, Hp, Wp, C) -> (B*16*twp, C) window-major padded rows (reference order vit.py:357-358)."""
    b, hp, wp, c = x_bhwc.shape
    h, w = hp // 4, wp // 4
    t = x_bhwc.reshape(b, 4, h, 4, w, c).permute(0, 1, 3, 2, 4, 5).reshape(b, 16, h * w, c)
    out = torch.zeros(b, 16, twp, c, dtype=x_bhwc.dtype, device=x_bhwc.device)
    out[:, :, :h * w] = t
    return out.reshape(b * 16 * twp, c)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hw", [(128, 192), (192, 192)])
def test_gemm_patch_embed_window_major(dtype, hw):
    from lwdetr_amd import kernels as K
    _ceil4 = lambda n: (n + 3) // 4 * 4
    hh, ww = hw
    b, c = 2, 192
    hp, wp = hh // 16, ww // 16
    twp = _ceil4((hp // 4) * (wp // 4))
    img = _rand(b, 3, hh, ww, dtype=dtype, seed=1)
    w = _rand(c, 3, 16, 16, dtype=dtype, scale=768 ** -0.5, seed=2)
    bias = _rand(c, seed=3)
    pos = _rand(16 * twp, c, dtype=dtype, seed=4)
    out = torch.zeros(b * 16 * twp, c, dtype=dtype, device=_dev())
    tok = K.tok_layout(True, hp, wp, twp)
    K.GemmOp(img, w.reshape(c, -1).contiguous(), b * 16 * twp, c, 768,
             [K.seg(out, 0, c, ldo=c, bias=bias, res=pos, ldres=c, res_mod=16 * twp)], a_mode=K.A_PATCH16, a_tok=tok,
             img_h=hh, img_w=ww)()
    ref = F.conv2d(img.float(), w.float(), bias, stride=16).permute(0, 2, 3, 1)
    ref = _to_winmajor(ref, twp).reshape(b, 16 * twp, c) + pos.float()
    valid = _to_winmajor(torch.ones(b, hp, wp, 1, device=_dev()), twp).reshape(b, 16 * twp, 1) > 0
    err = ((out.float().reshape(b, 16 * twp, c) - ref) * valid).abs().max() / ref.abs().max()
    assert err.item() < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("stride,winmajor", [(1, False), (2, True), (1, True), (2, False)])
def test_gemm_conv3x3(dtype, stride, winmajor):
    from lwdetr_amd import kernels as K
    b, hp, wp, cin, cout, ctot, col0 = 2, 12, 16, 64, 96, 160, 32
    x = _rand(b, hp, wp, ctot, dtype=dtype, seed=1)
    w = _rand(cout, cin, 3, 3, dtype=dtype, scale=(9 * cin) ** -0.5, seed=2)
    bias = _rand(cout, seed=3)
    twp = (hp // 4) * (wp // 4)
    a = _to_winmajor(x, twp) if winmajor else x.reshape(-1, ctot)
    ho, wo = (hp - 1) // stride + 1, (wp - 1) // stride + 1
    out = torch.zeros(b * ho * wo, cout, dtype=dtype, device=_dev())
    K.GemmOp(a.contiguous(), w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous(), b * ho * wo, cout, 9 * cin,
             [K.seg(out, 0, cout, ldo=cout, bias=bias, act=K.ACT_SILU)], lda=ctot, a_mode=K.A_CONV3x3,
             a_tok=K.tok_layout(winmajor, hp, wp, twp), conv_cin=cin, conv_stride=stride, a_col0=col0, conv_hout=ho,
             conv_wout=wo)()
    xin = x[..., col0:col0 + cin].float().permute(0, 3, 1, 2)
    ref = F.silu(F.conv2d(xin, w.float(), bias, stride=stride, padding=1)).permute(0, 2, 3, 1).reshape(-1, cout)
    assert _relerr(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("b,hp,wp,cin,cout,stride", [(1, 40, 40, 128, 128, 1), (2, 40, 40, 128, 128, 1), (1, 40, 40, 192, 192, 1), (1, 25, 40, 128, 256, 1),
                                                      (1, 40, 40, 128, 128, 2), (3, 24, 32, 192, 64, 2)])
def test_gemm_few_rows_conv3x3_and_plain(dtype, b, hp, wp, cin, cout, stride):
    """lwdetr_gemm_few (round 6: the few-row kernel - fragment-major weights straight from L2 into MFMA fragments, a third of the contraction per
    L2 round trip, no LDS ring) vs F.conv2d / the fp32 product and vs lwdetr_gemm on the same operands: the projector's 3x3 convolutions at one and
    two 640 x 640 images (40 x 40 pixels, 128 channels inside 640-wide rows), Cin = 192, a ragged last row tile (25 x 40 = 1000 = 62.5 tiles),
    stride 2, more / fewer output columns than one workgroup's 128; and the PLAIN view with bias + GELU + LayerScale + residual + tap copy."""
    from lwdetr_amd import kernels as K
    ctot, col0 = cin + 96, 32
    x = _rand(b, hp, wp, ctot, dtype=dtype, seed=1)
    w = _rand(cout, cin, 3, 3, dtype=dtype, scale=(9 * cin) ** -0.5, seed=2)
    bias = _rand(cout, seed=3)
    ho, wo = (hp - 1) // stride + 1, (wp - 1) // stride + 1
    m = b * ho * wo
    wk = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    assert K.gemm_few_supported(dtype, m, K.A_CONV3x3, cin)
    outs = []
    for cls, wt in ((K.GemmFewOp, K.pack_frag16(wk)), (K.GemmOp, wk)):
        out = torch.full((m, cout + 8), 7.0, dtype=dtype, device=_dev())
        cls(x.reshape(-1, ctot), wt, m, cout, 9 * cin, [K.seg(out, 0, cout, ldo=cout + 8, bias=bias, act=K.ACT_SILU)], lda=ctot, a_mode=K.A_CONV3x3,
            a_tok=K.tok_layout(False, hp, wp, 0), conv_cin=cin, conv_stride=stride, a_col0=col0, conv_hout=ho, conv_wout=wo)()
        outs.append(out)
    xin = x[..., col0:col0 + cin].float().permute(0, 3, 1, 2)
    ref = F.silu(F.conv2d(xin, w.float(), bias, stride=stride, padding=1)).permute(0, 2, 3, 1).reshape(-1, cout)
    assert _relerr(outs[0][:, :cout], ref) < TOL[dtype]
    assert bool((outs[0][:, cout:] == 7.0).all())
    assert _relerr(outs[0][:, :cout], outs[1][:, :cout].float()) < TOL[dtype] / 2
    # PLAIN view, the full LINEAR epilogue
    k, n = 9 * cin, cout
    a = _rand(m, k, dtype=dtype, seed=4)
    wl = _rand(n, k, dtype=dtype, scale=k ** -0.5, seed=5)
    gamma, res = _rand(n, seed=6), _rand(m, n, dtype=dtype, seed=7)
    o1, taps = torch.zeros(m, n, dtype=dtype, device=_dev()), torch.zeros(m, 2 * n, dtype=dtype, device=_dev())
    K.GemmFewOp(a, K.pack_frag16(wl), m, n, k, [K.seg(o1, 0, n, ldo=n, bias=bias, act=K.ACT_GELU, scale=0.5, gamma=gamma, res=res, ldres=n,
                                                   out2=taps[:, n:], ld2=2 * n)])()
    refp = res.float() + gamma * 0.5 * F.gelu(a.float() @ wl.float().t() + bias)
    assert _relerr(o1, refp) < TOL[dtype] and torch.equal(taps[:, n:], o1) and bool((taps[:, :n] == 0).all())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("stride,winmajor,cout", [(1, False, 192), (2, True, 384), (1, True, 128), (2, False, 256)])
@pytest.mark.parametrize("wg2", ["0", "2"])
def test_gemm_conv3x3_large_tile(dtype, stride, winmajor, cout, wg2, big_gemm, monkeypatch, knobs):
    """The implicit-GEMM 3x3 view on the 256-row large-tile kernel (column tiles 192 / 128 / 256), zero padding at the image
    border, stride 1 | 2, raster and window-major inputs, a channel window inside wider rows - vs F.conv2d. wg2 = 2: the 4-wave
    128-row form of the kernel (column tiles 256 / 192) wherever it is legal."""
    from lwdetr_amd import kernels as K
    if wg2 == "2":
        _need_experiments()
    knobs.set("GEMM_BIG_2WG", wg2)
    b, hp, wp, cin, ctot, col0 = 3, 24, 32, 128, 320, 64
    x = _rand(b, hp, wp, ctot, dtype=dtype, seed=1)
    w = _rand(cout, cin, 3, 3, dtype=dtype, scale=(9 * cin) ** -0.5, seed=2)
    bias = _rand(cout, seed=3)
    twp = (hp // 4) * (wp // 4)
    a = _to_winmajor(x, twp) if winmajor else x.reshape(-1, ctot)
    ho, wo = (hp - 1) // stride + 1, (wp - 1) // stride + 1
    out = torch.zeros(b * ho * wo, cout, dtype=dtype, device=_dev())
    K.GemmOp(a.contiguous(), w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous(), b * ho * wo, cout, 9 * cin,
             [K.seg(out, 0, cout, ldo=cout, bias=bias, act=K.ACT_SILU)], lda=ctot, a_mode=K.A_CONV3x3,
             a_tok=K.tok_layout(winmajor, hp, wp, twp), conv_cin=cin, conv_stride=stride, a_col0=col0, conv_hout=ho,
             conv_wout=wo)()
    xin = x[..., col0:col0 + cin].float().permute(0, 3, 1, 2)
    ref = F.silu(F.conv2d(xin, w.float(), bias, stride=stride, padding=1)).permute(0, 2, 3, 1).reshape(-1, cout)
    assert _relerr(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("b,hp,wp,c", [(2, 40, 40, 128), (3, 10, 13, 128), (2, 20, 20, 192), (1, 80, 80, 192), (5, 7, 9, 192), (1, 60, 60, 128)])
def test_gemm_conv3x3_patch_resident(dtype, b, hp, wp, c, monkeypatch, knobs):
    """The patch-resident 3x3 convolution (stride 1, raster rows, N = Cin: the C2f bottleneck convolutions) vs F.conv2d in fp64
    and vs the implicit-GEMM ring kernel it replaces: image borders, 128-pixel tiles that straddle images (10 x 13, 7 x 9), a
    ragged last tile, a channel window inside wider rows on both sides (the C2f concat buffer), the widest patch that fits."""
    from lwdetr_amd import kernels as K
    ctot, col0, ocol = 5 * c, 2 * c, 3 * c
    x = _rand(b, hp, wp, ctot, dtype=dtype, seed=1)
    w = _rand(c, c, 3, 3, dtype=dtype, scale=(9 * c) ** -0.5, seed=2)
    bias = _rand(c, seed=3)
    outs = []
    for patch in ("2", "0"):
        knobs.set("CONV_PATCH", patch)
        out = torch.full((b * hp * wp, ctot), 3.0, dtype=dtype, device=_dev())
        K.GemmOp(x.reshape(-1, ctot), w.permute(0, 2, 3, 1).reshape(c, -1).contiguous(), b * hp * wp, c, 9 * c,
                 [K.seg(out[:, ocol:], 0, c, ldo=ctot, bias=bias, act=K.ACT_SILU)], lda=ctot, a_mode=K.A_CONV3x3,
                 a_tok=K.tok_layout(False, hp, wp, 0), conv_cin=c, conv_stride=1, a_col0=col0, conv_hout=hp, conv_wout=wp,
                 keep=(out,))()
        torch.cuda.synchronize()
        assert bool((out[:, :ocol] == 3.0).all()) and bool((out[:, ocol + c:] == 3.0).all())        # nothing outside the segment
        outs.append(out[:, ocol:ocol + c].clone())
    xin = x[..., col0:col0 + c].double().permute(0, 3, 1, 2)
    ref = F.silu(F.conv2d(xin, w.double(), bias.double(), padding=1)).permute(0, 2, 3, 1).reshape(-1, c)
    e_new, e_old = _relerr(outs[0], ref.float()), _relerr(outs[1], ref.float())
    assert e_new < TOL[dtype] and e_new < 1.5 * e_old + 1e-4, (e_new, e_old)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_deconv2x2_and_tokmap(dtype):
    from lwdetr_amd import kernels as K
    b, hp, wp, cin, cout = 2, 8, 12, 64, 32
    twp = (hp // 4) * (wp // 4)
    x = _rand(b, hp, wp, cin, dtype=dtype, seed=1)
    a = _to_winmajor(x, twp)
    win = K.tok_layout(True, hp, wp, twp)
    # transposed conv 2x2 stride 2 with pixel shuffle into a wider concat buffer
    w = _rand(cin, cout, 2, 2, dtype=dtype, scale=cin ** -0.5, seed=2)
    bias = _rand(cout, seed=3)
    ntot, off = 3 * cout, cout
    out = torch.zeros(b * 4 * hp * wp, ntot, dtype=dtype, device=_dev())
    K.GemmOp(a, w.permute(2, 3, 1, 0).reshape(4 * cout, cin).contiguous(), a.shape[0], 4 * cout, cin,
             [K.seg(out[:, off:], 0, 4 * cout, mode=K.OUT_DECONV2x2, ldo=ntot, bias=bias.repeat(4).contiguous(), p0=cout,
                    in_tok=win, out_tok=K.tok_layout(False, 2 * hp, 2 * wp, 0), out_batch_stride=4 * hp * wp * ntot)])()
    ref = F.conv_transpose2d(x.float().permute(0, 3, 1, 2), w.float(), bias, stride=2).permute(0, 2, 3, 1)
    assert _relerr(out[:, off:off + cout], ref.reshape(-1, cout)) < TOL[dtype]
    # TOKMAP: window-major rows -> raster rows at a row offset inside a (B, S, n) buffer
    n, s_total, row_off = 64, hp * wp + 7, 7
    wl = _rand(n, cin, dtype=dtype, scale=cin ** -0.5, seed=4)
    o2 = torch.zeros(b * s_total, n, dtype=dtype, device=_dev())
    K.GemmOp(a, wl, a.shape[0], n, cin, [K.seg(o2, 0, n, mode=K.OUT_TOKMAP, ldo=n, in_tok=win,
                                               out_tok=K.tok_layout(False, hp, wp, 0), out_batch_stride=s_total * n,
                                               out_row_offset=row_off)])()
    ref2 = (x.float().reshape(b, hp * wp, cin) @ wl.float().t())
    assert _relerr(o2.reshape(b, s_total, n)[:, row_off:], ref2) < TOL[dtype]


def _attn_ref(q, k, v, valid):
    s = (q.float() @ k.float().transpose(-2, -1))
    s = s.masked_fill(~valid[None, None, None, :], float("-inf"))
    return s.softmax(-1) @ v.float()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hd", [16, 32, 64])
@pytest.mark.parametrize("geom", ["window100", "global1600", "holes", "decoder300", "holes3648", "global704", "window100_slack",
                                  "window228_slack"])
def test_attention(dtype, hd, geom):
    _attention_case(dtype, hd, geom, 0)


# every compiled shape of the LDS-ring kernel (the defaults are 104 at hd 16, 108 at hd 32, 208 at hd 64): waves per workgroup that do
# / do not divide the sequence, 64 queries per wave, 128 keys per ring step (U = 2: the second block of the last step lies past the sequence)
@pytest.mark.parametrize("cfg", [102, 103, 105, 106, 108, 110, 204, 205, 208, 1104, 1105, 1108, 1110, 1208])
@pytest.mark.parametrize("hd,dtype,geom", [(16, torch.float16, "global1600"), (32, torch.bfloat16, "holes3648"), (16, torch.bfloat16, "global704"),
                                           (32, torch.float16, "window228_slack"), (64, torch.float16, "holes3648")])
def test_attention_lds_ring_shapes(cfg, hd, dtype, geom):
    if hd == 64 and cfg >= 1000:
        pytest.skip("two 64-key blocks per ring step do not fit the LDS budget at hd 64")
    _attention_case(dtype, hd, geom, cfg)


def _attention_case(dtype, hd, geom, cfg):
    from lwdetr_amd import kernels as K
    heads, b = 3, 2
    if geom == "holes3648":        # the real 960x960 geometry (LDS-ring kernel for the 16-bit types): 225 tokens in 228 rows
        twp, tw, spi, b, heads = 228, 225, 1, 1, 2
    elif geom == "global704":      # 11 x 64 keys: partially filled last workgroup of the LDS-ring kernel
        twp, tw, spi = 44, 44, 1
    elif geom == "window228_slack":  # 960x960 windows (225 tokens in 228 rows) on the LDS-ring kernel (ragged last stage)
        twp, tw, spi, b, heads = 228, 225, 16, 1, 2
    elif geom in ("window100", "window100_slack"):
        twp, tw, spi = 100, 100, 16
    elif geom == "global1600":
        twp, tw, spi = 100, 100, 1
    elif geom == "holes":          # 15x15 windows padded to 228 rows (960x960 geometry), scaled down: 3x3 -> 12
        twp, tw, spi = 12, 9, 1
    else:
        twp, tw, spi = 300, 300, 1
    tp = 16 * twp if geom != "decoder300" else 300
    q = _rand(b, heads, tp, hd, dtype=dtype, seed=1)
    k = _rand(b, heads, tp, hd, dtype=dtype, seed=2)
    v = _rand(b, heads, tp, hd, dtype=dtype, seed=3)
    # spike one key against one query so the online-softmax rescale branch is exercised late in the sequence
    k[0, 0, tp - 3] = q[0, 0, 5] * 4
    k[0, 1, tp // 2 + 1] = q[0, 1, 40] * 4
    scale = K.attention_scale(hd)
    qs = (q.float() * scale).to(dtype)
    out = torch.zeros(b * tp, heads * hd, dtype=dtype, device=_dev())
    keys = twp if spi == 16 else tp
    slack = geom.endswith("_slack")             # V^T with 16 readable bytes behind it: lets the LDS-ring kernel take windows
    vt_store = torch.zeros(v.numel() + 8, dtype=dtype, device=_dev())
    vt = vt_store[:v.numel()].view(b, heads, hd, tp)
    vt.copy_(v.transpose(2, 3))
    from lwdetr_amd import _native
    # LDS-ring kernel for every 16-bit case it can serve (by default it only takes grids that fill the chip and >= 192 keys;
    # these test batches are small); the other geometries / f32 exercise attn_kernel
    use_lds = slack or geom in ("global1600", "holes3648", "global704")
    _native.lib().lwdetr_attention_tuning(3 if use_lds else -1)
    _native.lib().lwdetr_attention_tuning_cfg(cfg)
    try:
        K.AttnOp(qs, k, vt, out, B=b, heads=heads, hd=hd, Tp=tp, ldo=heads * hd,
                 seqs_per_img=spi, seq_tok_stride=twp if spi == 16 else tp, keys_per_seq=keys, sub_stride=twp,
                 sub_len=tw, kind=0, vt_slack=slack)()
    finally:
        _native.lib().lwdetr_attention_tuning(-1)
        _native.lib().lwdetr_attention_tuning_cfg(0)
    o = out.reshape(b, tp, heads, hd).permute(0, 2, 1, 3).float()
    qn = qs.float() / math.log2(math.e)      # kernel works in the log2 domain
    valid = (torch.arange(tp, device=_dev()) % twp) < tw
    if spi == 16:
        refs = []
        for wi in range(16):
            sl = slice(wi * twp, (wi + 1) * twp)
            refs.append(_attn_ref(qn[:, :, sl], k[:, :, sl], v[:, :, sl], valid[sl]))
        ref = torch.cat(refs, 2)
    else:
        ref = _attn_ref(qn, k, v, valid)
    err = ((o - ref)[:, :, valid]).abs().max().item()
    assert err < {torch.float32: 2e-5, torch.float16: 6e-3, torch.bfloat16: 4e-2}[dtype], err


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("hd", [16, 32])
@pytest.mark.parametrize("twp,tw,spi,b,heads", [(100, 100, 16, 2, 3), (28, 25, 16, 1, 3), (128, 126, 3, 1, 3), (36, 36, 16, 1, 2), (8, 5, 5, 1, 1),
                                                 (64, 64, 16, 1, 4), (100, 97, 16, 1, 2)])
def test_attention_one_wave_per_window(dtype, hd, twp, tw, spi, b, heads, monkeypatch, knobs):
    """attn_win_kernel (sequences of <= 128 keys, one wave per (sequence, head)) vs the fp32 formulation and vs attn_kernel:
    pad rows behind the real tokens, ragged key / query tiles, a partially filled last workgroup, 8 keys."""
    from lwdetr_amd import kernels as K
    tp = spi * twp
    q = _rand(b, heads, tp, hd, dtype=dtype, seed=1)
    k = _rand(b, heads, tp, hd, dtype=dtype, seed=2)
    v = _rand(b, heads, tp, hd, dtype=dtype, seed=3)
    k[0, 0, tw - 1] = q[0, 0, 2] * 4                  # a dominant key in the ragged last tile
    qs = (q.float() * K.attention_scale(hd)).to(dtype)
    vt = v.transpose(2, 3).contiguous()
    outs = []
    knobs.set("ATTN_WTILE", "0")
    for win in ("1", "0"):
        knobs.set("ATTN_WIN", win)
        out = torch.zeros(b * tp, heads * hd, dtype=dtype, device=_dev())
        K.AttnOp(qs, k, vt, out, B=b, heads=heads, hd=hd, Tp=tp, ldo=heads * hd, seqs_per_img=spi, seq_tok_stride=twp,
                 keys_per_seq=twp, sub_stride=twp, sub_len=tw, kind=0)()
        outs.append(out.reshape(b, tp, heads, hd).permute(0, 2, 1, 3).float())
    qn = qs.float() / math.log2(math.e)
    valid = (torch.arange(tp, device=_dev()) % twp) < tw
    ref = torch.cat([_attn_ref(qn[:, :, wi * twp:(wi + 1) * twp], k[:, :, wi * twp:(wi + 1) * twp], v[:, :, wi * twp:(wi + 1) * twp],
                               valid[wi * twp:(wi + 1) * twp]) for wi in range(spi)], 2)
    e_new = ((outs[0] - ref)[:, :, valid]).abs().max().item()
    e_old = ((outs[1] - ref)[:, :, valid]).abs().max().item()
    assert e_new < {torch.float16: 6e-3, torch.bfloat16: 4e-2}[dtype] and e_new < 1.5 * e_old + 1e-4, (e_new, e_old)
    assert torch.isfinite(outs[0]).all()              # pad rows are written too (finite)
    from lwdetr_amd import _native
    if hd == 16 and _native.lib().lwdetr_has_experiments():
        # round 5: the window-tile kernel (one workgroup per (image, window), V^T and the output tile through LDS) runs the same MFMA
        # sequence per (window, head) as the one-wave kernel: bit-identical, pad rows included
        knobs.set("ATTN_WTILE", "1")
        out = torch.zeros(b * tp, heads * hd, dtype=dtype, device=_dev())
        K.AttnOp(qs, k, vt, out, B=b, heads=heads, hd=hd, Tp=tp, ldo=heads * hd, seqs_per_img=spi, seq_tok_stride=twp,
                 keys_per_seq=twp, sub_stride=twp, sub_len=tw, kind=0)()
        ow = out.reshape(b, tp, heads, hd).permute(0, 2, 1, 3).float()
        assert torch.equal(ow, outs[0]), (ow - outs[0]).abs().max().item()
        # ... also into a wider output row (ldo > C) and with 12 heads (three per wave)
        b2, h2 = 3, 12
        q2, k2, v2 = (_rand(b2, h2, tp, hd, dtype=dtype, seed=s_) for s_ in (11, 12, 13))
        vt2 = v2.transpose(2, 3).contiguous()
        res = []
        for wt in ("1", "0"):
            knobs.set("ATTN_WTILE", wt)
            o2 = torch.full((b2 * tp, h2 * hd + 64), 7.0, dtype=dtype, device=_dev())
            K.AttnOp(q2, k2, vt2, o2, B=b2, heads=h2, hd=hd, Tp=tp, ldo=h2 * hd + 64, seqs_per_img=spi, seq_tok_stride=twp,
                     keys_per_seq=twp, sub_stride=twp, sub_len=tw, kind=0)()
            res.append(o2)
        assert bool((res[0][:, h2 * hd:] == 7.0).all())                       # nothing outside the rows' C channels
        val = ((torch.arange(tp, device=_dev()) % twp) < tw).repeat(b2)
        d2 = (res[0][:, :h2 * hd].float() - res[1][:, :h2 * hd].float())[val].abs().max().item()
        assert d2 < {torch.float16: 2e-3, torch.bfloat16: 2e-2}[dtype], d2


@pytest.fixture
def big_gemm():
    """Route every legal GEMM through the 256-row large-tile kernel (by default only large shapes take it)."""
    from lwdetr_amd import _native
    _native.lib().lwdetr_gemm_tuning(2)
    yield
    _native.lib().lwdetr_gemm_tuning(-1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mnk,depth", [((1000, 768, 768), 2), ((4099, 256, 3072), 2), ((2048, 3072, 768), 64), ((777, 384, 384), 32),
                                       ((513, 128, 1536), 2), ((300, 640, 1024), 32), ((700, 576, 512), 2), ((1000, 192, 448), 2),
                                       ((900, 1152, 384), 2),
                                       # round 5: the 4-wave / 128-row form (two workgroups per CU; column tiles 256 and 192), ragged M / N tails
                                       ((1000, 768, 768), 128), ((4099, 256, 3072), 128), ((2000, 2304, 768), 128), ((777, 192, 384), 128),
                                       ((129, 1100, 448), 128), ((5000, 3072, 768), 128)])
def test_gemm_large_tile_kernel(dtype, mnk, depth):
    """gemm_big_kernel (256 x 256 / 256 x 128 tiles, 32x32x16 MFMA, DMA ring of 32- and 64-deep stages; depth 128 = the 4-wave
    128 x 256 / 128 x 192 form) vs torch: bias, GELU, LayerScale + residual epilogue; ragged M and N tails."""
    from lwdetr_amd import _native, kernels as K
    if depth == 128:
        _need_experiments()
    m, n, k = mnk
    x = _rand(m, k, dtype=dtype, seed=1)
    w = _rand(n, k, dtype=dtype, scale=k ** -0.5, seed=2)
    bias = _rand(n, dtype=torch.float32, seed=3)
    gamma = _rand(n, dtype=torch.float32, seed=4)
    res = _rand(m, n, dtype=dtype, seed=5)
    _native.lib().lwdetr_gemm_tuning(depth)
    try:
        out = K.linear(x, w, bias, act=K.ACT_GELU, res=res, gamma=gamma)
        plain = K.linear(x, w, bias)
    finally:
        _native.lib().lwdetr_gemm_tuning(-1)
    base = K.linear(x, w, bias)                              # default kernel choice (64 x 64 tiles at these sizes)
    y = x.float() @ w.float().t() + bias
    ref = res.float() + gamma * torch.nn.functional.gelu(y)
    assert _relerr(out, ref) < TOL[dtype]
    assert _relerr(plain, y) < TOL[dtype]
    assert _relerr(plain, base.float()) < TOL[dtype]


@pytest.fixture
def pt_gemm():
    """Route every legal GEMM through the persistent large-tile kernel (gemm_pt.hip; by default only >= 16384-row shapes take it)."""
    from lwdetr_amd import _native
    _native.lib().lwdetr_gemm_pt_tuning(2)
    yield
    _native.lib().lwdetr_gemm_pt_tuning(-1)


def _pt_vs_big(run):
    """run() once through the persistent kernel (asserting that it served the launch) and once through gemm_big_kernel; returns both result lists."""
    from lwdetr_amd import _native
    lib = _native.lib()
    lib.lwdetr_gemm_tuning(2)
    try:
        lib.lwdetr_gemm_pt_tuning(2)
        n0 = lib.lwdetr_gemm_pt_count()
        a = run()
        torch.cuda.synchronize()
        assert lib.lwdetr_gemm_pt_count() > n0, "the persistent kernel refused a shape this test is meant to run on it"
        lib.lwdetr_gemm_pt_tuning(0)
        n1 = lib.lwdetr_gemm_pt_count()
        b = run()
        torch.cuda.synchronize()
        assert lib.lwdetr_gemm_pt_count() == n1
    finally:
        lib.lwdetr_gemm_pt_tuning(-1)
        lib.lwdetr_gemm_tuning(-1)
    return a, b


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mnk", [(4096, 768, 768),        # 48 tiles: one per workgroup, most workgroups idle
                                 (1000, 2304, 768),       # ragged last row tile (1000 = 3 x 256 + 232), 36 tiles
                                 (8192, 3072, 768),       # 384 tiles: one or two per workgroup
                                 (14592, 2304, 768),      # 513 tiles: two or three per workgroup (a quarter of xlarge's QKV)
                                 (20000, 768, 192),       # nk = 3: the ring parity flips from tile to tile; ragged tail
                                 (16392, 512, 128),       # nk = 2, the shortest contraction the kernel takes; 130 tiles, ragged by 8 rows
                                 (6144, 768, 3072)])      # fc2 of the C = 768 model: 48 steps per tile
def test_gemm_persistent_tile_kernel(dtype, mnk):
    """gemm_pt_kernel (round 6: persistent workgroups, DMA ring running across tile boundaries, register-direct epilogue) vs torch and
    vs gemm_big_kernel: bias + GELU + LayerScale + residual, plain bias, residual updated IN PLACE + tap copy (the ViT's fc2)."""
    from lwdetr_amd import kernels as K
    m, n, k = mnk
    x = _rand(m, k, dtype=dtype, seed=1)
    w = _rand(n, k, dtype=dtype, scale=k ** -0.5, seed=2)
    bias = _rand(n, dtype=torch.float32, seed=3)
    gamma = _rand(n, dtype=torch.float32, seed=4)
    res = _rand(m, n, dtype=dtype, seed=5)

    def run():
        out = K.linear(x, w, bias, act=K.ACT_GELU, res=res, gamma=gamma)
        plain = K.linear(x, w, bias)
        silu = K.linear(x, w, bias, act=K.ACT_SILU)
        inplace = res.clone()
        taps = torch.full((m, n + 64), 7.0, dtype=dtype, device=_dev())
        K.GemmOp(x, w, m, n, k, [K.seg(inplace, 0, n, ldo=n, bias=bias, gamma=gamma, res=inplace, ldres=n, out2=taps[:, 32:], ld2=n + 64)])()
        return [out, plain, silu, inplace, taps]

    pt, big = _pt_vs_big(run)
    y = x.float() @ w.float().t() + bias
    assert _relerr(pt[0], res.float() + gamma * F.gelu(y)) < TOL[dtype]
    assert _relerr(pt[1], y) < TOL[dtype]
    assert _relerr(pt[2], F.silu(y)) < TOL[dtype]
    assert _relerr(pt[3], res.float() + gamma * y) < TOL[dtype]
    assert torch.equal(pt[4][:, 32:32 + n], pt[3]) and bool((pt[4][:, :32] == 7.0).all()) and bool((pt[4][:, 32 + n:] == 7.0).all())
    # against gemm_big_kernel: the same products summed from the bias instead of from zero, one scale multiplication instead of two -
    # equal up to an f32 rounding or two in front of the 16-bit rounding (a few outputs land on the neighbouring 16-bit value)
    for a, b, what in zip(pt, big, ("gelu + layerscale + residual", "bias", "silu", "in place", "taps")):
        assert _relerr(a, b) < TOL[dtype] / 2, (what, _relerr(a, b))
        assert (a != b).float().mean().item() < 0.02, (what, (a != b).float().mean().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("b,tp", [(4, 1600), (3, 1000), (2, 3648)])
def test_gemm_persistent_tile_kernel_head_layouts(dtype, b, tp):
    """QKV of the C = 768 model through the persistent kernel: HEADS (q with bias and scale, k) and HEADS_T (V^T: swapped MFMA operands, 8 consecutive
    tokens per lane); tiles that straddle an image boundary at a non-aligned row (tp = 1000) and a ragged last tile; also against gemm_big_kernel."""
    from lwdetr_amd import kernels as K
    heads, hd = 12, 64
    c = heads * hd
    x = _rand(b * tp, c, dtype=dtype, seed=1)
    w = _rand(3 * c, c, dtype=dtype, scale=c ** -0.5, seed=2)
    qb, vb = _rand(c, seed=3), _rand(c, seed=4)

    def run():
        q = torch.zeros(b, heads, tp, hd, dtype=dtype, device=_dev())
        k = torch.zeros_like(q)
        vt = torch.zeros(b, heads, hd, tp, dtype=dtype, device=_dev())
        K.GemmOp(x, w, b * tp, 3 * c, c, [
            K.seg(q, 0, c, mode=K.OUT_HEADS, bias=qb, scale=0.37, p0=tp, p1=hd, p2=heads),
            K.seg(k, c, 2 * c, mode=K.OUT_HEADS, p0=tp, p1=hd, p2=heads),
            K.seg(vt, 2 * c, 3 * c, mode=K.OUT_HEADS_T, bias=vb, p0=tp, p1=hd, p2=heads)])()
        return [q, k, vt]

    pt, big = _pt_vs_big(run)
    y = x.float() @ w.float().t()
    sp = lambda t: t.reshape(b, tp, heads, hd).permute(0, 2, 1, 3)
    assert _relerr(pt[0], sp((y[:, :c] + qb) * 0.37)) < TOL[dtype]
    assert _relerr(pt[1], sp(y[:, c:2 * c])) < TOL[dtype]
    assert _relerr(pt[2], sp(y[:, 2 * c:] + vb).transpose(2, 3)) < TOL[dtype]
    for a, b_, what in zip(pt, big, "q k vt".split()):
        assert _relerr(a, b_) < TOL[dtype] / 2, (what, _relerr(a, b_))
        assert (a != b_).float().mean().item() < 0.02, (what, (a != b_).float().mean().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("hd", [64, 32, 16])
@pytest.mark.parametrize("wg2", ["0", "2"])
def test_gemm_large_tile_kernel_head_layouts(dtype, hd, wg2, big_gemm, monkeypatch, knobs):
    """QKV of the C = 768 / 384 / 192 models through the large-tile kernel (column tiles 256 / 192 / 192; wg2 = 2: its 4-wave
    128-row form): HEADS (q, k) and HEADS_T (V^T, swapped MFMA operands)."""
    if wg2 == "2":
        _need_experiments()
    knobs.set("GEMM_BIG_2WG", wg2)
    _check_qkv_layouts(dtype, hd, 4, 1600)
    _check_qkv_layouts(dtype, hd, 1, 1000)          # ragged last row tile (1000 = 7 x 128 + 104)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("m,c,mode", [(1000, 768, 2), (4160, 768, 64), (2048, 256, 2), (4099, 768, 2), (16384, 768, -1), (700, 1024, 2)])
def test_gemm_with_layernorm_folded_in(dtype, m, c, mode):
    """Round 5: LayerNorm folded into the following GEMM (lwdetr_gemm_seg.ln_stats / ln_colsum + lwdetr_row_stats + kernels.fold_layernorm)
    vs the torch fp32 formulation LN(x) W^T + b and vs the two launches it replaces (lwdetr_layernorm + GEMM): QKV-shaped (HEADS / HEADS /
    HEADS_T segments, scaled q) and fc1-shaped (GELU) outputs, rows with a large common offset (mean >> std: the cancellation case),
    ragged M. The fold lives in the 256 x 256 large-tile kernel's own epilogue (gemm_big_ln_kernel): mode 2 / 64 force that kernel, -1 at
    16 384 rows takes it by itself; shapes it does not serve are refused (last lines)."""
    _need_experiments()
    from lwdetr_amd import _native, kernels as K
    heads = 12 if c % 12 == 0 else 8
    hd = c // heads
    x = (_rand(m, c, dtype=torch.float32, seed=1) * (0.5 + 2 * _rand(m, 1, dtype=torch.float32, seed=2).abs()) + 8 * _rand(m, 1, dtype=torch.float32, seed=3)).to(dtype)
    lw, lb = _rand(c, seed=4) * 0.2 + 1, _rand(c, seed=5) * 0.1
    wqkv = _rand(3 * c, c, scale=c ** -0.5, seed=6)
    bqkv = torch.cat([_rand(c, seed=7) * 0.1, torch.zeros(c, device=_dev()), _rand(c, seed=8) * 0.1])
    w1, b1 = _rand(4 * c, c, scale=c ** -0.5, seed=9), _rand(4 * c, seed=10) * 0.1
    stats = torch.empty(2, m, device=_dev())                # planar: row 0 = mean, row 1 = rstd
    K.RowStatsOp(x, stats, m, c, 1e-6)()
    xf = x.float()
    assert (stats[0] - xf.mean(1)).abs().max().item() < 1e-4 * (1 + xf.abs().max().item())
    rs = (xf.var(1, unbiased=False) + 1e-6).rsqrt()
    assert ((stats[1] - rs).abs() / rs).max().item() < 1e-4
    ln = F.layer_norm(xf, (c,), lw, lb, 1e-6)
    tp = m
    sp = lambda t_: t_.reshape(1, tp, heads, hd).permute(0, 2, 1, 3)
    _native.lib().lwdetr_gemm_tuning(mode)
    try:
        # QKV
        wq_, cs_, bq_ = K.fold_layernorm(wqkv, bqkv, lw, lb, dtype)
        if m % 4 == 0:
            q = torch.zeros(1, heads, tp, hd, dtype=dtype, device=_dev()); k = torch.zeros_like(q)
            vt = torch.zeros(1, heads, hd, tp, dtype=dtype, device=_dev())
            K.GemmOp(x, wq_, m, 3 * c, c, [
                K.seg(q, 0, c, mode=K.OUT_HEADS, bias=bq_[:c], scale=0.37, p0=tp, p1=hd, p2=heads, ln_stats=stats, ln_colsum=cs_[:c]),
                K.seg(k, c, 2 * c, mode=K.OUT_HEADS, bias=bq_[c:2 * c], p0=tp, p1=hd, p2=heads, ln_stats=stats, ln_colsum=cs_[c:2 * c]),
                K.seg(vt, 2 * c, 3 * c, mode=K.OUT_HEADS_T, bias=bq_[2 * c:], p0=tp, p1=hd, p2=heads, ln_stats=stats, ln_colsum=cs_[2 * c:])])()
            y = ln @ wqkv.t() + bqkv
            tol = TOL[dtype] * 2
            assert _relerr(q, sp(y[:, :c]) * 0.37) < tol and _relerr(k, sp(y[:, c:2 * c])) < tol
            assert _relerr(vt, sp(y[:, 2 * c:]).transpose(2, 3)) < tol
        # fc1 + GELU, against the fp32 formulation and against the two launches it replaces
        w1_, cs1_, b1_ = K.fold_layernorm(w1, b1, lw, lb, dtype)
        hid = torch.zeros(m, 4 * c, dtype=dtype, device=_dev())
        K.GemmOp(x, w1_, m, 4 * c, c, [K.seg(hid, 0, 4 * c, ldo=4 * c, bias=b1_, act=K.ACT_GELU, ln_stats=stats, ln_colsum=cs1_)])()
        ref = F.gelu(ln @ w1.t() + b1)
        two = K.linear(K.layernorm(x, lw, lb, 1e-6), w1.to(dtype), b1, act=K.ACT_GELU)
        e_new, e_old = _relerr(hid, ref), _relerr(two, ref)
        assert e_new < TOL[dtype] * 2 and e_new < 1.5 * e_old + 1e-4, (e_new, e_old)
    finally:
        _native.lib().lwdetr_gemm_tuning(-1)
    # a shape the large-tile kernel does not take (few rows, default thresholds) is refused, not silently computed without the LayerNorm
    if m < 16384:
        with pytest.raises(_native.NativeError):
            K.GemmOp(x, w1_, m, 4 * c, c, [K.seg(hid, 0, 4 * c, ldo=4 * c, bias=b1_, act=K.ACT_GELU, ln_stats=stats, ln_colsum=cs1_)])()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("splits", [2, 3, 4])
def test_gemm_split_k_few_rows(dtype, splits):
    """Round 5: split-K of the 64 x 64 DMA ring kernel for few-row GEMMs with a long contraction (lwdetr_gemm_desc.splitk / splitk_ws: slices of
    a tile write f32 slabs, the last arriver sums them in slice order and runs the epilogue) - the single-image shapes: the projector's 3x3
    convolution (implicit GEMM, SiLU), a 1x1 convolution into a TOKMAP destination shape, a decoder Linear with residual + ragged M / N.
    Against torch fp32 and against the unsplit launch; repeated launches are bit-identical (the sum does not depend on who arrives last) and
    leave the arrival counters at zero."""
    _need_experiments()
    from lwdetr_amd import kernels as K
    # (a) 3x3 convolution, 40 x 40 pixels, 128 -> 128 channels inside wider rows
    b, hp, wp, c, ctot, col0 = 1, 40, 40, 128, 256, 64
    x = _rand(b, hp, wp, ctot, dtype=dtype, seed=1)
    w = _rand(c, c, 3, 3, dtype=dtype, scale=(9 * c) ** -0.5, seed=2)
    bias = _rand(c, seed=3)
    outs = []
    for sk in (splits, 0, splits):
        out = torch.zeros(b * hp * wp, c, dtype=dtype, device=_dev())
        op = K.GemmOp(x.reshape(-1, ctot), w.permute(0, 2, 3, 1).reshape(c, -1).contiguous(), b * hp * wp, c, 9 * c,
                      [K.seg(out, 0, c, ldo=c, bias=bias, act=K.ACT_SILU)], lda=ctot, a_mode=K.A_CONV3x3, a_tok=K.tok_layout(False, hp, wp, 0),
                      conv_cin=c, conv_stride=1, a_col0=col0, conv_hout=hp, conv_wout=wp, splitk=sk)
        assert op.desc.splitk == (sk if sk >= 2 else 0)
        for _ in range(3):
            op()
        torch.cuda.synchronize()
        if sk >= 2:
            ws = op._keep[4]
            tiles = ((b * hp * wp + 63) // 64) * ((c + 63) // 64)
            assert ws[tiles * sk * K.SPLITK_SLAB:].abs().max().item() == 0          # counters back at zero
        outs.append(out)
    ref = F.silu(F.conv2d(x[..., col0:col0 + c].float().permute(0, 3, 1, 2), w.float(), bias, padding=1)).permute(0, 2, 3, 1).reshape(-1, c)
    assert _relerr(outs[0], ref) < TOL[dtype] and _relerr(outs[1], ref) < TOL[dtype]
    assert torch.equal(outs[0], outs[2])                                              # bit-reproducible
    assert _relerr(outs[0], outs[1].float()) < TOL[dtype] / 2
    # (b) plain GEMMs: ragged M and N, residual + LayerScale, two segments
    for (m, n, k) in [(1600, 256, 768), (300, 256, 512), (1000, 200, 1024), (77, 64, 2048)]:
        xa = _rand(m, k, dtype=dtype, seed=4)
        wa = _rand(n, k, dtype=dtype, scale=k ** -0.5, seed=5)
        ba, ga = _rand(n, seed=6), _rand(n, seed=7)
        res = _rand(m, n, dtype=dtype, seed=8)
        got = []
        for sk in (splits, 0):
            o = torch.zeros(m, n, dtype=dtype, device=_dev())
            K.GemmOp(xa, wa, m, n, k, [K.seg(o, 0, n, ldo=n, bias=ba, gamma=ga, res=res, ldres=n, act=K.ACT_RELU)], splitk=sk)()
            got.append(o)
        refp = res.float() + ga * torch.relu(xa.float() @ wa.float().t() + ba)
        assert _relerr(got[0], refp) < TOL[dtype], (m, n, k)
        assert _relerr(got[0], got[1].float()) < TOL[dtype] / 2, (m, n, k)
    # the automatic policy: off (measured slower on the latency path: kernels.splitk_for)
    assert K.splitk_for(1600, 128, 1152, dtype) == 1 and K.splitk_for(51200, 128, 1152, dtype) == 1


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c", [192, 256, 384, 768])
def test_layernorm(dtype, c):
    from lwdetr_amd import kernels as K
    m = 1003
    x = _rand(m, c, dtype=dtype, seed=1) * 3 + 0.5
    g, b = _rand(c, seed=2), _rand(c, seed=3)
    out = K.layernorm(x, g, b, 1e-6)
    ref = F.layer_norm(x.float(), (c,), g, b, 1e-6)
    assert (out.float() - ref).abs().max().item() < {torch.float32: 2e-5, torch.float16: 8e-3, torch.bfloat16: 6e-2}[dtype]
    # batched row remap (projector LayerNorm writing into a level slice of memory)
    rows, s_total, off = 50, 64, 9
    xb = _rand(2 * rows, c, dtype=dtype, seed=4)
    ob = torch.zeros(2 * s_total, c, dtype=dtype, device=_dev())
    K.LayerNormOp(xb, g, b, ob, 2 * rows, c, 1e-6, rows_per_batch=rows, out_batch_rows=s_total, out_row_offset=off)()
    refb = F.layer_norm(xb.float(), (c,), g, b, 1e-6).reshape(2, rows, c)
    assert (ob.reshape(2, s_total, c)[:, off:off + rows].float() - refb).abs().max().item() < 6e-2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c", [256, 384])
def test_layernorm_chain_is_bit_identical_to_two_launches(dtype, c):
    """norm3 + decoder.norm in one launch (lwdetr_layernorm_chain) == two lwdetr_layernorm launches, bit for bit."""
    from lwdetr_amd import kernels as K
    m = 9600
    x = _rand(m, c, dtype=dtype, seed=1) * 3
    g1, b1, g2, b2 = (_rand(c, seed=s_) for s_ in (2, 3, 4, 5))
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    K.LayerNormChainOp(x, g1, b1, 1e-5, o1, g2, b2, 1e-6, o2, m, c)()
    r1 = K.layernorm(x, g1, b1, 1e-5)
    r2 = K.layernorm(r1, g2, b2, 1e-6)
    assert torch.equal(o1, r1) and torch.equal(o2, r2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("c,hid,m", [(256, 2048, 9600), (256, 2048, 300), (256, 2048, 19200), (384, 2048, 4800), (384, 2048, 301),
                                     (256, 1024, 77), (256, 2048, 40000), (256, 64, 100)])
def test_ffn_fused(dtype, c, hid, m):
    """Decoder FFN (split-hidden partial products + finishing LayerNorm chain, two launches) vs the torch fp32 formulation
    of transformer.py:507-512 / :397-400 and vs the unfused three-launch plan. 9600 / 19200 = 32 / 64 images x 300 queries
    (BASELINE configs 2-4), 300 = one image, 40000 = more tiles than one workgroup per CU takes, 64 = a single chunk pair."""
    from lwdetr_amd import kernels as K
    assert K.ffn_fused_supported(c, hid, dtype)
    x = _rand(m, c, dtype=dtype, seed=1)
    w1, b1 = _rand(hid, c, scale=c ** -0.5, seed=2), _rand(hid, seed=3) * 0.1
    w2, b2 = _rand(c, hid, scale=hid ** -0.5, seed=4), _rand(c, seed=5) * 0.1
    g1, be1 = _rand(c, seed=6) * 0.2 + 1, _rand(c, seed=7) * 0.1
    g2, be2 = _rand(c, seed=8) * 0.2 + 1, _rand(c, seed=9) * 0.1
    xf = x.float()
    r1 = F.layer_norm(xf + F.relu(xf @ w1.t() + b1) @ w2.t() + b2, (c,), g1, be1, 1e-5)
    r2 = F.layer_norm(r1, (c,), g2, be2, 1e-5)
    w1p, b1p, w2c = K.pack_mlp_weights(w1, b1, w2, None, None, dtype)
    assert torch.equal(w1p, w1.to(dtype)) and torch.equal(b1p, b1)
    o1, o2 = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
    op = K.FfnOp(x, w1p, b1p, w2c, b2, g1, be1, 1e-5, o1, g2, be2, 1e-5, o2, m, c)
    assert op.splits >= 1 and (hid // 32) % op.splits == 0
    op()
    tol = {torch.float16: 4e-3, torch.bfloat16: 3e-2}[dtype]
    assert _relerr(o1, r1) < tol, _relerr(o1, r1)
    assert _relerr(o2, r2) < tol, _relerr(o2, r2)
    # the unfused plan (two GEMM launches + LayerNorm chain) on the same 16-bit weights: same roundings up to f32 summation order
    h = K.linear(x, w1.to(dtype), b1, act=K.ACT_RELU)
    y = K.linear(h, w2.to(dtype), b2, res=x)
    u1 = K.layernorm(y, g1, be1, 1e-5)
    assert _relerr(o1, u1) < tol / 2, _relerr(o1, u1)
    # in place (the engine's use: out1 aliases x), and deterministic
    xx = x.clone()
    o3 = torch.empty_like(x)
    K.FfnOp(xx, w1p, b1p, w2c, b2, g1, be1, 1e-5, xx, g2, be2, 1e-5, o3, m, c)()
    assert torch.equal(xx, o1) and torch.equal(o3, o2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c,m", [(192, 1000), (192, 64), (192, 6400), (384, 333), (192, 51200), (192, 20000), (192, 64000), (384, 25600)])
def test_mlp_fused(dtype, c, m):
    """LN -> fc1 -> GELU -> fc2 -> LayerScale -> residual in one launch vs the unfused torch fp32 formulation.
    The large M are the BASELINE full sizes (51200 = 32 images x 1600 tokens: 6-7 tiles per workgroup, the LDS-resident
    residual path), a ragged tile count (20000) and more than 7 tiles per CU (64000: extra rounds); 64 / 1000 rows run the
    few-token kernel with 16-token workgroups, 6400 rows (4 images) with 32-token ones."""
    from lwdetr_amd import kernels as K
    if not K.mlp_fused_supported(c, dtype):
        pytest.skip("not instantiated for this shape / dtype")
    x = _rand(m, c, dtype=dtype, seed=1) * 2 + 0.3
    w1, b1 = _rand(4 * c, c, scale=c ** -0.5, seed=2), _rand(4 * c, seed=3) * 0.1
    w2, b2 = _rand(c, 4 * c, scale=(4 * c) ** -0.5, seed=4), _rand(c, seed=5) * 0.1
    lw, lb, g2 = _rand(c, seed=6) * 0.2 + 1, _rand(c, seed=7) * 0.1, _rand(c, seed=8) * 0.3
    xf = x.float()
    ref = xf + g2 * (F.gelu(F.layer_norm(xf, (c,), lw, lb, 1e-6) @ w1.t() + b1) @ w2.t() + b2)
    w1f, b1f, w2c = K.pack_mlp_weights(w1, b1, w2, lw, lb, dtype)
    out2 = torch.zeros(m, 2 * c, dtype=dtype, device=_dev())
    stats = torch.zeros(m, 2, device=_dev())
    xx = x.clone()
    K.MlpFusedOp(xx, w1f, b1f, w2c, b2, g2, m, c, 1e-6, out2=out2[:, c:], ld2=2 * c, stats_out=stats, eps_next=1e-6)()
    tol = {torch.float32: 3e-5, torch.float16: 6e-3, torch.bfloat16: 5e-2}[dtype]
    assert _relerr(xx, ref) < tol, _relerr(xx, ref)
    assert torch.equal(out2[:, c:], xx) and out2[:, :c].abs().max().item() == 0
    mean, var = xx.float().mean(1), xx.float().var(1, unbiased=False)
    assert (stats[:, 0] - mean).abs().max().item() < 1e-4
    assert ((stats[:, 1] - (var + 1e-6).rsqrt()).abs() / (var + 1e-6).rsqrt()).max().item() < 1e-4
    # with the attention output projection fused in front: x1 = x + g1 * (att Wp^T + bp), then the MLP on x1
    att = _rand(m, c, dtype=dtype, seed=9)
    wp, bp, g1 = _rand(c, c, scale=c ** -0.5, seed=10), _rand(c, seed=11) * 0.1, _rand(c, seed=12) * 0.3
    x1 = xf + g1 * (att.float() @ wp.t() + bp)
    x1r = x1.to(dtype).float()                      # the kernel rounds x1 to the storage type before the MLP
    ref2 = x1r + g2 * (F.gelu(F.layer_norm(x1r, (c,), lw, lb, 1e-6) @ w1.t() + b1) @ w2.t() + b2)
    w1p, b1p, w2p = K.pack_mlp_weights(w1, b1, w2, lw, lb, dtype, proj=True)
    xx = x.clone()
    K.MlpFusedOp(xx, w1p, b1p, w2p, b2, g2, m, c, 1e-6, att=att, wp=wp.to(dtype).contiguous(), bp=bp, gamma1=g1)()
    assert _relerr(xx, ref2) < tol, _relerr(xx, ref2)
    # ... and with the next block's LayerNorm + QKV chained behind it (Q / K head layout, V transposed)
    heads, hd = 12, c // 12
    tp = {1000: 100, 64: 64, 6400: 1600, 333: None, 51200: 1600, 20000: 400, 64000: 1600, 25600: 1600}[m]
    if tp is not None:
        nb = m // tp
        wqkv = _rand(3 * c, c, scale=c ** -0.5, seed=13)
        qb, vb = _rand(c, seed=14) * 0.1, _rand(c, seed=15) * 0.1
        lw1, lb1 = _rand(c, seed=16) * 0.2 + 1, _rand(c, seed=17) * 0.1
        wq, bq = K.pack_qkv_weights(wqkv, qb, vb, lw1, lb1, dtype)
        q = torch.zeros(nb, heads, tp, hd, dtype=dtype, device=_dev())
        k = torch.zeros_like(q)
        vt = torch.zeros(nb, heads, hd, tp, dtype=dtype, device=_dev())
        xx2 = x.clone()
        K.MlpFusedOp(xx2, w1p, b1p, w2p, b2, g2, m, c, 1e-6, att=att, wp=wp.to(dtype).contiguous(), bp=bp, gamma1=g1,
                     wqkv=wq, bqkv=bq, q=q, k=k, vt=vt, qscale=0.37, heads=heads, hd=hd, Tp=tp)()
        assert torch.equal(xx2, xx)
        y = F.layer_norm(xx2.float(), (c,), lw1, lb1, 1e-6) @ wqkv.t() + torch.cat([qb, torch.zeros_like(qb), vb])
        sp = lambda t_: t_.reshape(nb, tp, heads, hd).permute(0, 2, 1, 3)
        assert _relerr(q, sp(y[:, :c]) * 0.37) < tol * 2
        assert _relerr(k, sp(y[:, c:2 * c])) < tol * 2
        assert _relerr(vt, sp(y[:, 2 * c:]).transpose(2, 3)) < tol * 2


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("m,tp", [(1600, 1600), (3200, 1600), (6400, 1600), (1000, 100), (2496, 624), (12000, 400)])
def test_vit_block_few_fragment_major_weights(dtype, m, tp):
    """lwdetr_vit_block_few (round 6: the few-token ViT block kernel on fragment-major Wp / W1 / Wqkv - pack_frag16) is the arithmetic of
    lwdetr_mlp_fused bit for bit: residual stream, tap copy, row statistics, chained q / k / v^T; also against the torch fp32 formulation.
    1600 / 3200 rows = one / two 640 x 640 images (16-token workgroups), the rest 32-token workgroups incl. ragged tile counts."""
    from lwdetr_amd import kernels as K
    c, heads = 192, 12
    hd = c // heads
    assert K.vit_block_few_supported(c, dtype, m)
    x = _rand(m, c, dtype=dtype, seed=1) * 2 + 0.3
    att = _rand(m, c, dtype=dtype, seed=9)
    w1, b1 = _rand(4 * c, c, scale=c ** -0.5, seed=2), _rand(4 * c, seed=3) * 0.1
    w2, b2 = _rand(c, 4 * c, scale=(4 * c) ** -0.5, seed=4), _rand(c, seed=5) * 0.1
    lw, lb = _rand(c, seed=6) * 0.2 + 1, _rand(c, seed=7) * 0.1
    g2, g1 = _rand(c, seed=8) * 0.3, _rand(c, seed=12) * 0.3
    wp, bp = _rand(c, c, scale=c ** -0.5, seed=10), _rand(c, seed=11) * 0.1
    wqkv = _rand(3 * c, c, scale=c ** -0.5, seed=13)
    qb, vb = _rand(c, seed=14) * 0.1, _rand(c, seed=15) * 0.1
    lw1, lb1 = _rand(c, seed=16) * 0.2 + 1, _rand(c, seed=17) * 0.1
    w1p, b1p, w2p = K.pack_mlp_weights(w1, b1, w2, lw, lb, dtype, proj=True)
    wq, bq = K.pack_qkv_weights(wqkv, qb, vb, lw1, lb1, dtype)
    wpd = wp.to(dtype).contiguous()
    nb = m // tp
    res = []
    for few in (False, True):
        xx = x.clone()
        taps = torch.full((m, 2 * c), 7.0, dtype=dtype, device=_dev())
        stats = torch.zeros(m, 2, device=_dev())
        q = torch.zeros(nb, heads, tp, hd, dtype=dtype, device=_dev())
        k = torch.zeros_like(q)
        vt = torch.zeros(nb, heads, hd, tp, dtype=dtype, device=_dev())
        cls = K.VitBlockFewOp if few else K.MlpFusedOp
        f = K.pack_frag16 if few else (lambda t: t)
        cls(xx, f(w1p), b1p, w2p, b2, g2, m, c, 1e-6, out2=taps[:, c:], ld2=2 * c, stats_out=stats, att=att, wp=f(wpd), bp=bp, gamma1=g1,
            wqkv=f(wq), bqkv=bq, q=q, k=k, vt=vt, qscale=0.37, heads=heads, hd=hd, Tp=tp)()
        res.append((xx, taps, stats, q, k, vt))
    for a, b, what in zip(res[1], res[0], "x taps stats q k vt".split()):
        assert torch.equal(a, b), (what, (a.float() - b.float()).abs().max().item())
    x1 = (x.float() + g1 * (att.float() @ wp.t() + bp)).to(dtype).float()
    ref = x1 + g2 * (F.gelu(F.layer_norm(x1, (c,), lw, lb, 1e-6) @ w1.t() + b1) @ w2.t() + b2)
    assert _relerr(res[1][0], ref) < {torch.float16: 6e-3, torch.bfloat16: 5e-2}[dtype]
    assert bool((res[1][1][:, :c] == 7.0).all())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("c,heads,m,tp", [(192, 6, 51200, 1600), (192, 12, 12800, 400), (192, 3, 20000, 400), (192, 6, 64000, 1600),
                                          (384, 12, 25600, 1600), (384, 12, 12816, 4272), (192, 6, 13000, 1000)])
@pytest.mark.parametrize("half", ["0", "1"])
@pytest.mark.parametrize("gelu16", ["0", "1"])
def test_vit_block(dtype, c, heads, m, tp, half, gelu16, monkeypatch, knobs):
    """gelu16 = 1: the GELU on packed f16 pairs (the f16 default since round 5, LWDETR_VB_GELU16=0|1, f16 only; tests/vitblock_sim.py:gelu_vb16_packed) - same bounds
    against the erf-GELU fp32 formulation, different bits from the f32-arithmetic form and within 2e-3 of it. half = 1: the C = 192 form with 32 tokens per wave / two workgroups per CU (round 5; the default while all workgroups of a launch
    are resident at once), half = 0: 64 tokens per wave. lwdetr_vit_block (attention projection + LayerScale + residual, norm2 -> fc1 -> GELU -> fc2 -> LayerScale -> residual, and
    norm1 + QKV of the next block, one launch) vs the torch fp32 formulation of vit.py:195-222 and vs lwdetr_mlp_fused on the
    same 16-bit weights. 51200 = BASELINE config 2 (32 images x 1600 tokens: 50 tokens per wave), 64000 / 25600 = more than one
    round of workgroups, 12816 / 13000 / 20000 = ragged token counts per wave and tiles that straddle images."""
    from lwdetr_amd import kernels as K
    if c != 192 and half == "1":
        pytest.skip("the half-tile form exists for C = 192 only")
    if gelu16 == "1" and dtype != torch.float16:
        pytest.skip("packed-f16 GELU: f16 only")
    knobs.set("VB_HALF", half)
    knobs.set("VB_GELU16", gelu16)
    hd = c // heads
    assert K.vit_block_supported(c, dtype, hd)
    x = _rand(m, c, dtype=dtype, seed=1) * 2 + 0.3
    att = _rand(m, c, dtype=dtype, seed=9)
    w1, b1 = _rand(4 * c, c, scale=c ** -0.5, seed=2), _rand(4 * c, seed=3) * 0.1
    w2, b2 = _rand(c, 4 * c, scale=(4 * c) ** -0.5, seed=4), _rand(c, seed=5) * 0.1
    lw, lb = _rand(c, seed=6) * 0.2 + 1, _rand(c, seed=7) * 0.1
    g2 = _rand(c, seed=8) * 0.3
    g2 = torch.where(g2.abs() < 0.02, torch.full_like(g2, 0.02), g2)
    wp, bp = _rand(c, c, scale=c ** -0.5, seed=10), _rand(c, seed=11) * 0.1
    g1 = _rand(c, seed=12) * 0.3
    g1 = torch.where(g1.abs() < 0.02, torch.full_like(g1, -0.02), g1)
    wqkv = _rand(3 * c, c, scale=c ** -0.5, seed=13)
    qb, vb = _rand(c, seed=14) * 0.1, _rand(c, seed=15) * 0.1
    lw1, lb1 = _rand(c, seed=16) * 0.2 + 1, _rand(c, seed=17) * 0.1
    xf = x.float()
    x1 = (xf + g1 * (att.float() @ wp.t() + bp)).to(dtype).float()          # the kernel rounds x1 to the storage type
    ref = x1 + g2 * (F.gelu(F.layer_norm(x1, (c,), lw, lb, 1e-6) @ w1.t() + b1) @ w2.t() + b2)
    tol = {torch.float16: 6e-3, torch.bfloat16: 5e-2}[dtype]
    nb = m // tp if m % tp == 0 else None
    for with_qkv in ([True, False] if nb else [False]):
        stream, vec = K.pack_vit_block(wp, bp, g1, w1, b1, w2, b2, g2, lw, lb, dtype,
                                       qkv=(wqkv, qb, vb, lw1, lb1) if with_qkv else None)
        stream, vec = stream.to(_dev()), vec.to(_dev())
        xx = x.clone()
        out2 = torch.zeros(m, 2 * c, dtype=dtype, device=_dev())
        stats = torch.zeros(m, 2, device=_dev())
        kw = {}
        if with_qkv:
            q = torch.full((nb, heads, tp, hd), float("nan"), dtype=dtype, device=_dev())
            k = torch.full_like(q, float("nan"))
            vt = torch.full((nb, heads, hd, tp), float("nan"), dtype=dtype, device=_dev())
            kw = dict(q=q, k=k, vt=vt, qscale=0.37, heads=heads, hd=hd, Tp=tp)
        K.VitBlockOp(xx, att, stream, vec, m, c, 1e-6, out2=out2[:, c:], ld2=2 * c, stats_out=stats, eps_next=1e-6, **kw)()
        torch.cuda.synchronize()
        assert torch.isfinite(xx.float()).all()
        assert _relerr(xx, ref) < tol, _relerr(xx, ref)
        assert torch.equal(out2[:, c:], xx) and out2[:, :c].abs().max().item() == 0
        mean, var = xx.float().mean(1), xx.float().var(1, unbiased=False)
        assert (stats[:, 0] - mean).abs().max().item() < 1e-4
        assert ((stats[:, 1] - (var + 1e-6).rsqrt()).abs() / (var + 1e-6).rsqrt()).max().item() < 1e-4
        if with_qkv:
            y = F.layer_norm(xx.float(), (c,), lw1, lb1, 1e-6) @ wqkv.t() + torch.cat([qb, torch.zeros_like(qb), vb])
            sp = lambda t_: t_.reshape(nb, tp, heads, hd).permute(0, 2, 1, 3)
            assert torch.isfinite(q.float()).all() and torch.isfinite(k.float()).all() and torch.isfinite(vt.float()).all()
            assert _relerr(q, sp(y[:, :c]) * 0.37) < tol * 2
            assert _relerr(k, sp(y[:, c:2 * c])) < tol * 2
            assert _relerr(vt, sp(y[:, 2 * c:]).transpose(2, 3)) < tol * 2
        # deterministic, and agrees with the round-2 kernel (same 16-bit weights, different summation order) well inside the bound
        xx2 = x.clone()
        K.VitBlockOp(xx2, att, stream, vec, m, c, 1e-6, **kw)()
        assert torch.equal(xx2, xx)
        if gelu16 == "1":                        # the switch is read per launch: the same op with the f32-arithmetic GELU
            knobs.set("VB_GELU16", "0")
            xx3 = x.clone()
            K.VitBlockOp(xx3, att, stream, vec, m, c, 1e-6, **kw)()
            knobs.set("VB_GELU16", "1")
            assert not torch.equal(xx3, xx) and _relerr(xx3, xx) < 2e-3, _relerr(xx3, xx)
    w1p, b1p, w2p = K.pack_mlp_weights(w1, b1, w2, lw, lb, dtype, proj=True)
    xo = x.clone()
    K.MlpFusedOp(xo, w1p, b1p, w2p, b2, g2, m, c, 1e-6, att=att, wp=wp.to(dtype).contiguous(), bp=bp, gamma1=g1)()
    assert _relerr(xx, xo) < tol / 2, _relerr(xx, xo)


@pytest.mark.parametrize("gelu", ["f32", "packed_f16"])
def test_vit_block_rounding_points_fp64(gelu, monkeypatch, knobs):
    """The 16-bit block kernel against an fp64 evaluation of the same arithmetic with the kernel's rounding points (16-bit
    inputs / weights, x1 and the output rounded to f16, f32 accumulation otherwise): the benchmarked kernel itself meets the
    1e-3 bar of the fp32 parity gate when its inputs are exactly representable. gelu = packed_f16: the f16
    default since round 5 (every GELU operation rounded to f16, tests/vitblock_sim.py:gelu_vb16_packed; the hardware's exp2 / rcp are
    within an ulp of the model's correctly rounded ones, hence the wider bounds)."""
    from lwdetr_amd import kernels as K
    from tests.vitblock_sim import gelu_vb16, gelu_vb16_packed
    knobs.set("VB_GELU16", "1" if gelu == "packed_f16" else "0")
    if gelu == "packed_f16":
        gelu_vb16 = gelu_vb16_packed                                    # noqa: F811
    c, m, dtype = 192, 12800, torch.float16
    r16 = lambda t: t.to(dtype).double()
    x, att = r16(_rand(m, c, seed=1) * 2 + 0.3), r16(_rand(m, c, seed=9))
    w1, b1 = _rand(4 * c, c, scale=c ** -0.5, seed=2).double(), (_rand(4 * c, seed=3) * 0.1).double()
    w2, b2 = _rand(c, 4 * c, scale=(4 * c) ** -0.5, seed=4).double(), (_rand(c, seed=5) * 0.1).double()
    lw, lb = torch.ones(c, dtype=torch.float64, device=_dev()), torch.zeros(c, dtype=torch.float64, device=_dev())
    g1, g2 = (_rand(c, seed=12) * 0.1 + 0.4).double(), (_rand(c, seed=8) * 0.1 + 0.3).double()
    wp, bp = _rand(c, c, scale=c ** -0.5, seed=10).double(), (_rand(c, seed=11) * 0.1).double()
    stream, vec = K.pack_vit_block(wp, bp, g1, w1, b1, w2, b2, g2, lw, lb, dtype)
    # the weights the kernel sees: 16-bit roundings of the (identity-LayerNorm-folded) f32 masters
    wp16, w116, w216 = r16(wp.float()), r16(w1.float()), r16(w2.float())
    x1 = r16((x + g1.float().double() * (att @ wp16.t() + bp.float().double())).float())
    ln = r16(((x1 - x1.mean(1, keepdim=True)) / (x1.var(1, unbiased=False, keepdim=True) + 1e-6).sqrt()).float())
    hid = ln @ w116.t() + b1.float().double()
    hid = r16(torch.from_numpy(gelu_vb16(hid.cpu().numpy())).to(_dev()).float())
    ref = r16((x1 + g2.float().double() * (hid @ w216.t() + b2.float().double())).float())
    xx = x.to(dtype).clone()
    K.VitBlockOp(xx, att.to(dtype), stream.to(_dev()), vec.to(_dev()), m, c, 1e-6)()
    d = (xx.double() - ref).abs()
    # one f16 ulp of the result at most on a few elements (f32 vs f64 accumulation order), far below 1e-3 relative on average
    k_ = 1 if gelu == "f32" else 3
    assert d.max().item() <= k_ * 2 * 2.0 ** -10 * ref.abs().max().item(), d.max().item()
    assert (d.mean() / ref.abs().mean()).item() < k_ * 1e-4, (d.mean() / ref.abs().mean()).item()


def test_gemm_rejects_bad_arguments():
    from lwdetr_amd import kernels as K
    from lwdetr_amd._native import NativeError
    x, w = _rand(8, 48), _rand(8, 48)
    with pytest.raises(NativeError):
        K.linear(x, w)                       # K not a multiple of 32
