"""GPU: the row-chain kernel (lwdetr_enc_chain, csrc/chain.hip) against a plain PyTorch fp32 reference of the same chain with the
unfused launches' rounding points: [cv2 1x1 conv + SiLU + LayerNorm ->] memory -> value projections (padding mask on output rows),
enc_output on rows with invalid proposals zeroed + LayerNorm, class logits + row maximum."""
import pytest
import torch

import lwdetr_amd  # noqa: F401
from lwdetr_amd import kernels as K

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ln(x, g, b, eps):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), g, b, eps)


@pytest.mark.parametrize("d,k5,dtype,B,npix", [(256, 640, torch.float16, 2, 1600), (256, 640, torch.bfloat16, 3, 100), (256, 0, torch.float16, 1, 333),
                                                (256, 0, torch.bfloat16, 2, 1600), (384, 0, torch.float16, 2, 2125), (384, 0, torch.bfloat16, 1, 77)])
def test_enc_chain_matches_torch(d, k5, dtype, B, npix):
    g = torch.Generator().manual_seed(d + k5 + npix)
    r = lambda *s: torch.randn(*s, generator=g)
    nl, ncls, S = 3, 91, npix
    M = B * npix
    w_enc, b_enc, g_enc, be_enc = r(d, d) / 16, r(d) * 0.5, 1 + 0.1 * r(d), 0.1 * r(d)
    w_cls, b_cls = r(ncls, d) / 16, r(ncls)
    w_val, b_val = r(nl * d, d) / 16, r(nl * d) * 0.5
    cv2 = (r(d, k5) / 25, r(d) * 0.5, 1 + 0.1 * r(d), 0.1 * r(d)) if k5 else None
    stream, vec = K.pack_enc_chain(d, dtype, w_enc, b_enc, g_enc, be_enc, w_cls, b_cls, w_val, b_val, cv2=cv2)
    x = (r(M, k5 or d) * (1.0 if k5 else 1.5)).to(dtype)
    rowvalid = (torch.rand(M, generator=g) > 0.3).to(torch.uint8)
    notpad = (torch.rand(M, generator=g) > 0.2).to(torch.uint8)
    # ---- reference: fp32 arithmetic on the 16-bit operands, every stage output rounded to the storage type
    T = lambda t: t.to(dtype).float()
    Wt = lambda w: w.to(dtype).float()
    xf = x.float()
    if k5:
        z = T(torch.nn.functional.silu(xf @ Wt(cv2[0]).T + cv2[1]))
        mem = T(_ln(z, cv2[2], cv2[3], 1e-6))
    else:
        mem = xf
    vals = T(mem @ Wt(w_val).T + b_val) * notpad[:, None].float()
    om = T(_ln(T((mem * rowvalid[:, None].float()) @ Wt(w_enc).T + b_enc), g_enc, be_enc, 1e-5))
    cls = T(om @ Wt(w_cls).T + b_cls)
    # ---- kernel
    dev = lambda t: t.to(DEV)
    memory = torch.full((M, d), 7.0, dtype=dtype, device=DEV)
    omo = torch.full((M, d), 7.0, dtype=dtype, device=DEV)
    clso = torch.full((M, 96), 7.0, dtype=dtype, device=DEV)
    cmax = torch.full((M,), 7.0, dtype=torch.float32, device=DEV)
    values = [torch.full((M, d), 7.0, dtype=dtype, device=DEV) for _ in range(nl)]
    xin = dev(x)
    op = K.EncChainOp(xin, k5 or d, k5, memory if k5 else None, omo, clso, 96, cmax, values, dev(rowvalid), dev(notpad), dev(stream), dev(vec),
                      M=M, d=d, npix=npix, S=S, lsi=0, total_rows=M, ncls=ncls, eps_p=1e-6, eps_e=1e-5)
    op()
    torch.cuda.synchronize()
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7

    def close(got, ref, what, k=3.0):
        got, ref = got.float().cpu(), ref
        err = (got - ref).abs()
        bound = k * ulp * ref.abs().clamp(min=0.25)
        frac = (err > bound).float().mean().item()
        assert frac < 2e-3 and err.max().item() < 16 * k * ulp * max(1.0, ref.abs().max().item()), (what, frac, err.max().item())

    if k5:
        close(memory, mem, "memory")
    for i in range(nl):
        close(values[i], vals[:, i * d:(i + 1) * d], f"values[{i}]")
        assert (values[i].float().cpu()[notpad == 0] == 0).all()
    close(omo, om, "om", k=4.0)
    close(clso[:, :ncls], cls, "cls", k=6.0)
    assert (clso[:, ncls:] == 0).all()
    # the row maximum is the maximum of the kernel's own (rounded) class logits, exactly
    assert torch.equal(cmax, clso[:, :ncls].float().max(1).values)


def _run_row_chain(d, dtype, k_in, M, stages, res=None, qpos=None, x=None):
    """stages as RowChainOp takes them (f32 masters, outputs allocated here) -> (op outputs dict, reference outputs dict)."""
    dev = lambda t: None if t is None else t.to(DEV)
    T = lambda t: t.to(dtype).float()
    outs, refs = {}, {}
    cur = x.float()
    for i, st in enumerate(stages):
        w16 = st["w"].to(dtype).float()
        if st["kind"] == "full":
            y = cur @ w16.T + st["b"]
            if st.get("res"):
                y = y + res.float()
            if st.get("relu"):
                y = y.relu()
            y = T(y)
            if st.get("ln") is not None:
                y = T(_ln(y, st["ln"][0], st["ln"][1], st["ln"][2]))
            if st.get("store"):
                st["out"] = torch.full((M, d), 7.0, dtype=dtype, device=DEV)
                outs[i], refs[i] = st["out"], y
            cur = T(y + qpos.float()) if st.get("addq") else y
        else:
            n = st["w"].shape[0]
            st["out"] = torch.full((M, st["ldo"]), 7.0, dtype=dtype, device=DEV)
            outs[i], refs[i] = st["out"], T(cur @ w16.T + st["b"])
    stream, vec = K.RowChainOp.pack(d, dtype, k_in, stages)
    op = K.RowChainOp(dev(x), x.shape[1], k_in, stages, dev(stream), dev(vec), M=M, d=d, res=dev(res), ld_res=d if res is not None else 0,
                      qpos=dev(qpos), ld_q=d if qpos is not None else 0)
    op._hold = (stream, vec)
    op()
    torch.cuda.synchronize()
    return outs, refs


@pytest.mark.parametrize("name,d,dtype,M", [("front", 256, torch.float16, 4800), ("front", 256, torch.bfloat16, 300), ("back", 256, torch.float16, 1111),
                                            ("heads", 256, torch.bfloat16, 900), ("heads", 384, torch.float16, 14400), ("bbox", 384, torch.bfloat16, 300),
                                            ("front", 384, torch.float16, 9600), ("back", 384, torch.bfloat16, 301), ("front", 384, torch.bfloat16, 33),
                                            ("refpoint", 256, torch.float16, 2500),
                                            # more than 16 384 rows: the row-per-wave form (fewer: 32-row workgroups, channels split over the waves)
                                            ("front", 256, torch.float16, 14400), ("front", 256, torch.float16, 19200), ("heads", 256, torch.bfloat16, 28800)])
def test_row_chain_matches_torch(name, d, dtype, M):
    g = torch.Generator().manual_seed(M + d)
    r = lambda *s: torch.randn(*s, generator=g)
    lin = lambda n, k: (r(n, k) / (k ** 0.5), r(n) * 0.3)
    lnp = lambda eps: (1 + 0.1 * r(d), 0.1 * r(d), eps)
    res = qpos = None
    k_in = d
    if name == "front":
        noa = 96 if d == 256 else 576                  # sampling_offsets | attention_weights: heads * levels * points * 3
        w, b = lin(d, d); wo, bo = lin(noa, d)
        res, qpos = r(M, d).to(dtype), r(M, d).to(dtype)
        stages = [dict(kind="full", w=w, b=b, res=True, ln=lnp(1e-5), store=True, addq=True), dict(kind="side", w=wo, b=bo, ldo=noa)]
    elif name == "back":
        w, b = lin(d, d)
        res = r(M, d).to(dtype)
        stages = [dict(kind="full", w=w, b=b, res=True, ln=lnp(1e-5), store=True)]
    elif name == "heads":
        wc, bc = lin(91, d); w0, b0 = lin(d, d); w1, b1 = lin(d, d); w2, b2 = lin(4, d)
        stages = [dict(kind="side", w=wc, b=bc, ldo=92), dict(kind="full", w=w0, b=b0, relu=True), dict(kind="full", w=w1, b=b1, relu=True),
                  dict(kind="side", w=w2, b=b2, ldo=4)]
    elif name == "bbox":
        w0, b0 = lin(d, d); w1, b1 = lin(d, d); w2, b2 = lin(4, d)
        stages = [dict(kind="full", w=w0, b=b0, relu=True), dict(kind="full", w=w1, b=b1, relu=True), dict(kind="side", w=w2, b=b2, ldo=4)]
    else:
        k_in = 2 * d
        w0, b0 = lin(d, k_in); w1, b1 = lin(d, d)
        stages = [dict(kind="full", w=w0, b=b0, relu=True), dict(kind="full", w=w1, b=b1, store=True)]
    x = r(M, k_in).to(dtype)
    outs, refs = _run_row_chain(d, dtype, k_in, M, stages, res=res, qpos=qpos, x=x)
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    assert outs
    for i in outs:
        got, ref = outs[i].float().cpu(), refs[i]
        n = ref.shape[1]
        err = (got[:, :n] - ref).abs()
        k = 3.0 + 2.0 * i                          # rounding differences compound along the chain
        bound = k * ulp * ref.abs().clamp(min=0.25)
        frac = (err > bound).float().mean().item()
        assert frac < 3e-3 and err.max().item() < 16 * k * ulp * max(1.0, ref.abs().max().item()), (name, i, frac, err.max().item())
        if got.shape[1] > (n + 3) // 4 * 4:
            assert (got[:, (n + 3) // 4 * 4:] == 7.0).all(), "columns beyond ceil4(n) must not be written"


@pytest.mark.parametrize("dtype,c,heads,m,tp", [(torch.float16, 192, 12, 51200, 1600), (torch.bfloat16, 192, 12, 12800, 1600), (torch.float16, 384, 12, 25600, 1600),
                                                (torch.bfloat16, 384, 12, 3648 * 4, 3648), (torch.float16, 192, 12, 1600, 1600)])
def test_vit_qkv_matches_torch(dtype, c, heads, m, tp):
    """lwdetr_vit_qkv (norm1 + QKV of block 0 in one launch) vs LayerNorm + Linear in fp32 on the same 16-bit operands (vit.py:199, :123-130)."""
    g = torch.Generator().manual_seed(c + m)
    r = lambda *s: torch.randn(*s, generator=g)
    hd = c // heads
    x = (r(m, c) * 2 + 0.3).to(dtype)
    wqkv, qb, vb = r(3 * c, c) / c ** 0.5, r(c) * 0.1, r(c) * 0.1
    lw, lb = r(c) * 0.2 + 1, r(c) * 0.1
    stream, vec = K.pack_vit_qkv(wqkv, qb, vb, lw, lb, dtype)
    nb = m // tp
    q = torch.full((nb, heads, tp, hd), float("nan"), dtype=dtype, device=DEV)
    k = torch.full_like(q, float("nan"))
    vt = torch.full((nb, heads, hd, tp), float("nan"), dtype=dtype, device=DEV)
    K.VitQkvOp(x.to(DEV), stream.to(DEV), vec.to(DEV), m, c, 1e-6, q=q, k=k, vt=vt, qscale=0.37, heads=heads, hd=hd, Tp=tp)()
    torch.cuda.synchronize()
    xn = torch.nn.functional.layer_norm(x.float(), (c,), lw, lb, 1e-6)
    ref = xn @ wqkv.t() + torch.cat([qb, torch.zeros(c), vb])
    rq = (ref[:, :c] * 0.37).view(nb, tp, heads, hd).permute(0, 2, 1, 3)
    rk = ref[:, c:2 * c].view(nb, tp, heads, hd).permute(0, 2, 1, 3)
    rv = ref[:, 2 * c:].view(nb, tp, heads, hd).permute(0, 2, 3, 1)
    tol = {torch.float16: 6e-3, torch.bfloat16: 5e-2}[dtype]
    for got, want, name in ((q, rq, "q"), (k, rk, "k"), (vt, rv, "vt")):
        got = got.float().cpu()
        assert torch.isfinite(got).all(), name
        err = (got - want).abs().max().item() / want.abs().max().item()
        assert err < tol, (name, err)


@pytest.mark.parametrize("dtype,c,heads,b,hp,twp", [(torch.float16, 192, 12, 8, 40, 100), (torch.bfloat16, 192, 12, 3, 40, 100), (torch.float16, 384, 12, 4, 40, 100),
                                                    (torch.bfloat16, 384, 12, 2, 60, 228), (torch.float16, 192, 12, 1, 44, 121)])
def test_vit_stem_matches_torch(dtype, c, heads, b, hp, twp):
    """lwdetr_vit_stem (patch embedding + position embedding + norm1 + QKV of block 0, one launch) vs Conv2d + add + LayerNorm + Linear in
    fp32 on the same 16-bit operands (vit.py:353-358, :199, :123-130); 960 x 960 geometry: 225-token windows in 228 rows (zero pad rows)."""
    g = torch.Generator().manual_seed(c + b + hp)
    r = lambda *s: torch.randn(*s, generator=g)
    hd, tp, hw = c // heads, 16 * twp, hp // 4
    img = r(b, 3, 16 * hp, 16 * hp).to(dtype)
    wpe, bpe = (r(c, 3, 16, 16) / 768 ** 0.5), r(c) * 0.1
    wqkv, qb, vb = r(3 * c, c) / c ** 0.5, r(c) * 0.1, r(c) * 0.1
    lw, lb = r(c) * 0.2 + 1, r(c) * 0.1
    pos = torch.zeros(16, twp, c)
    pos[:, :hw * hw] = r(16, hw * hw, c) * 0.5
    pos = pos.reshape(tp, c).to(dtype)
    stream, vec = K.pack_vit_stem(wpe, bpe, wqkv, qb, vb, lw, lb, dtype)
    m = b * tp
    x = torch.full((m, c), float("nan"), dtype=dtype, device=DEV)
    q = torch.full((b, heads, tp, hd), float("nan"), dtype=dtype, device=DEV)
    k = torch.full_like(q, float("nan"))
    vt = torch.full((b, heads, hd, tp), float("nan"), dtype=dtype, device=DEV)
    op = K.VitStemOp(img.to(DEV), pos.to(DEV), x, stream.to(DEV), vec.to(DEV), b, hp, hp, twp, c, 1e-6, q=q, k=k, vt=vt, qscale=0.37, heads=heads, hd=hd)
    op()
    torch.cuda.synchronize()
    wd = wpe.to(dtype).float()
    y = torch.nn.functional.conv2d(img.float(), wd, bpe, stride=16).permute(0, 2, 3, 1)                  # (b, hp, hp, c) raster
    y = y.reshape(b, 4, hw, 4, hw, c).permute(0, 1, 3, 2, 4, 5).reshape(b, 16, hw * hw, c)                 # window-major
    x0 = torch.zeros(b, 16, twp, c)
    x0[:, :, :hw * hw] = y
    x0[:, :, hw * hw:] = bpe                                                                               # pad rows: zero pixels
    x0 = (x0.reshape(b, tp, c) + pos.float()[None]).to(dtype)
    valid = (torch.arange(tp) % twp) < hw * hw
    gx = x.float().cpu().reshape(b, tp, c)
    assert torch.isfinite(gx).all()
    tolx = {torch.float16: 3e-3, torch.bfloat16: 2.5e-2}[dtype]
    errx = (gx - x0.float())[:, valid].abs().max().item() / x0.float().abs().max().item()
    assert errx < tolx, ("x0", errx)
    xn = torch.nn.functional.layer_norm(gx, (c,), lw, lb, 1e-6)                                           # from the kernel's own rounded x0
    ref = xn @ (wqkv * 1.0).t() + torch.cat([qb, torch.zeros(c), vb])
    rq = (ref[..., :c] * 0.37).view(b, tp, heads, hd).permute(0, 2, 1, 3)
    rk = ref[..., c:2 * c].view(b, tp, heads, hd).permute(0, 2, 1, 3)
    rv = ref[..., 2 * c:].view(b, tp, heads, hd).permute(0, 2, 3, 1)
    tol = {torch.float16: 6e-3, torch.bfloat16: 5e-2}[dtype]
    for got, want, name, tdim in ((q, rq, "q", 2), (k, rk, "k", 2), (vt, rv, "vt", 3)):
        got = got.float().cpu()
        assert torch.isfinite(got).all(), name
        d = (got - want).abs()
        d = d[:, :, valid] if tdim == 2 else d[..., valid]
        err = d.max().item() / want.abs().max().item()
        assert err < tol, (name, err)
