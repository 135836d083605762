"""Lane-level emulation of lw-detr_amd/csrc/vitblock.hip on the CPU (test infrastructure, numpy).

It walks the PACKED weight stream exactly as the kernel does - fragment f of piece p is the 1 KB at (p * KS + f) * 1024 bytes,
lane l owns elements [8 l, 8 l + 8) of it - and applies the gfx950 32x32x16 MFMA register layouts, so the host-side packing
(lwdetr_amd.kernels.pack_vit_block) and the kernel's index arithmetic are checked against the dense formulation without a GPU.
Arithmetic is float64 (no 16-bit rounding): this validates layouts, not numerics."""
import numpy as np


def mfma_32x32x16(a_frag, b_frag, c):
    """a_frag, b_frag (64, 8): lane (i | j = l & 31, h = l >> 5) holds A[i][8h + s] / B[8h + s][j]; c (64, 16): register 4 b + e of
    lane (j, h) is D[8 b + 4 h + e][j]. Returns D in the same register layout."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for lane in range(64):
        i, h = lane & 31, lane >> 5
        A[i, 8 * h:8 * h + 8] = a_frag[lane]
        B[8 * h:8 * h + 8, i] = b_frag[lane]
    D = A @ B
    out = np.array(c, dtype=np.float64, copy=True)
    for lane in range(64):
        j, h = lane & 31, lane >> 5
        for r in range(16):
            out[lane, r] += D[8 * (r // 4) + 4 * h + (r % 4), j]
    return out


def gelu_fast16(x):
    x2 = np.minimum(x * x, 36.0)
    p = (x2 * 0.0010142630555 - 0.1067757240036) * x2 - 2.3011213394584
    return x / (1.0 + np.exp2(x * p))


def gelu_vb16(x):
    """The block kernel's two-term form (vitblock.hip:vb_gelu16)."""
    return x / (1.0 + np.exp2(x * (-2.3087653 - 0.10012561 * x * x)))


def gelu_vb16_packed(x):
    """The block kernel's opt-in packed-f16 form (vitblock.hip: VB_G16_*, LWDETR_VB_GELU16=1): the same two-term expression with EVERY
    operation rounded to f16 - x itself first (v_cvt_pk_f16_f32), then v_pk_mul (x^2), v_pk_fma (c1 x^2 + c0, one rounding), v_pk_mul,
    v_exp_f16, v_pk_add, v_rcp_f16, v_pk_mul. exp2 / rcp are modelled correctly rounded (the hardware's are within an ulp)."""
    h = lambda v: np.asarray(v, dtype=np.float64).astype(np.float16).astype(np.float64)
    c0 = float(np.array([0xc09e], dtype=np.uint16).view(np.float16)[0]); c1 = float(np.array([0xae68], dtype=np.uint16).view(np.float16)[0])
    with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
        xh = h(x)
        t = h(xh * h(h(xh * xh) * c1 + c0))
        return h(xh * h(1.0 / h(1.0 + h(np.exp2(t)))))


def simulate_wave(stream, vec, x, att, t0, nvalid, C, NH, eps, eps_next, qkv=None, order="pipelined"):
    """One wave of the kernel: tokens [t0, t0 + nvalid) of x / att (M, C). Returns (new rows (nvalid, C), dict of q/k/v writes).
    qkv = dict(heads, hd, Tp, qscale) or None; writes are returned as {("q"|"k"|"v", flat element index): value}."""
    KS, NTI, NCH = C // 16, C // 32, C // 8
    stream = np.asarray(stream, dtype=np.float64).reshape(-1, KS, 64, 8)          # piece, fragment, lane, element
    vec = np.asarray(vec, dtype=np.float64)
    b1s, bps, g1s, rg1s = vec[:4 * C], vec[4 * C:5 * C], vec[5 * C:6 * C], vec[6 * C:7 * C]
    b2s, rg2s, g2s, bqs = vec[7 * C:8 * C], vec[8 * C:9 * C], vec[9 * C:10 * C], vec[10 * C:13 * C]
    lanes = np.arange(64); J, H = lanes & 31, lanes >> 5
    H0, Q0 = NTI, NTI + 2 * NCH

    def row(mat, th):           # token rows of this lane (zeros past nvalid: buffer loads return 0 out of range)
        tok = 32 * th + J
        r = np.zeros((64, C))
        ok = tok < nvalid
        r[ok] = mat[t0 + tok[ok]]
        return r

    # attention rows as B fragments (natural k order), x rows in accumulator layout
    xf = [[None] * KS for _ in range(NH)]
    for th in range(NH):
        a = row(att, th)
        for t in range(KS):
            xf[th][t] = np.stack([a[l, 16 * t + 8 * H[l]:16 * t + 8 * H[l] + 8] for l in range(64)])
    acc2 = [[np.zeros((64, 16)) for _ in range(NH)] for _ in range(NTI)]
    for th in range(NH):
        xr = row(x, th)
        for n in range(NTI):
            for b in range(4):
                for e in range(4):
                    c0 = 32 * n + 8 * b + 4 * H + e
                    acc2[n][th][:, 4 * b + e] = xr[lanes, c0] * rg1s[c0] + bps[c0]
    for n in range(NTI):
        for t in range(KS):
            for th in range(NH):
                acc2[n][th] = mfma_32x32x16(stream[n, t], xf[th][t], acc2[n][th])
    # x1, LayerNorm, B fragments, fc2 accumulator start
    for th in range(NH):
        x1 = np.zeros((64, NTI, 16))
        for n in range(NTI):
            for r in range(16):
                c0 = 32 * n + 8 * (r // 4) + 4 * H + (r % 4)
                x1[:, n, r] = g1s[c0] * acc2[n][th][:, r]
        s = x1.sum(axis=(1, 2)); s = s + s[lanes ^ 32]
        mean = s / C
        v = ((x1 - mean[:, None, None]) ** 2).sum(axis=(1, 2)); v = v + v[lanes ^ 32]
        rstd = 1.0 / np.sqrt(v / C + eps)
        for n in range(NTI):
            for be in range(2):
                f = np.zeros((64, 8))
                for s8 in range(8):
                    f[:, s8] = (x1[:, n, (2 * be + (s8 >> 2)) * 4 + (s8 & 3)] - mean) * rstd
                xf[th][2 * n + be] = f
            for r in range(16):
                c0 = 32 * n + 8 * (r // 4) + 4 * H + (r % 4)
                acc2[n][th][:, r] = x1[:, n, r] * rg2s[c0] + b2s[c0]

    def bias16(src):
        out = np.zeros((64, 16))
        for r in range(16):
            out[:, r] = src[8 * (r // 4) + 4 * H + (r % 4)]
        return out

    def fc1(piece, k):
        acc = [bias16(b1s[32 * k:32 * k + 32]) for _ in range(NH)]
        for t in range(KS):
            for th in range(NH):
                acc[th] = mfma_32x32x16(stream[piece, t], xf[th][t], acc[th])
        return acc

    def gelu_to_hf(acc):
        hf = [[np.zeros((64, 8)) for _ in range(2)] for _ in range(NH)]
        for th in range(NH):
            y = gelu_fast16(acc[th])
            for r in range(16):
                bq, e = r // 4, r % 4
                hf[th][bq >> 1][:, 4 * (bq & 1) + e] = y[:, r]
        return hf

    def fc2(piece, hf):
        for fi in range(2 * NTI):
            kap, n = fi // NTI, fi % NTI
            for th in range(NH):
                acc2[n][th] = mfma_32x32x16(stream[piece, fi], hf[th][kap], acc2[n][th])

    acc1 = {0: fc1(H0, 0)}
    hfs = {}
    if order == "pipelined":        # the 4-wave kernel: iteration k = GELU(k) beside fc2(k-1) and fc1(k+1)
        for k in range(NCH):
            hfs[k] = gelu_to_hf(acc1[k])
            if k >= 1:
                fc2(H0 + 2 * k, hfs[k - 1])                      # W2c(k-1)
            if k + 1 < NCH:
                acc1[k + 1] = fc1(H0 + 2 * k + 1, k + 1)         # W1c(k+1)
        fc2(H0 + 2 * NCH - 1, hfs[NCH - 1])
    else:                           # the 8-wave kernel: GELU(k) task, then the MFMA task fc2(k) + fc1(k+1); pairs, 2 pad pieces
        for k in range(NCH):
            hfs[k] = gelu_to_hf(acc1[k])
            fc2(H0 + 2 + 2 * k, hfs[k])                          # W2c(k): first piece of pair 4 + k
            if k + 1 < NCH:
                acc1[k + 1] = fc1(H0 + 3 + 2 * k, k + 1)         # W1c(k+1): second piece
        Q0 += 2

    out = np.zeros((nvalid, C))
    writes = {}
    for th in range(NH):
        xn = np.zeros((64, NTI, 16))
        for n in range(NTI):
            for r in range(16):
                c0 = 32 * n + 8 * (r // 4) + 4 * H + (r % 4)
                xn[:, n, r] = g2s[c0] * acc2[n][th][:, r]
                tok = 32 * th + J
                ok = tok < nvalid
                out[tok[ok], c0[ok]] = xn[ok, n, r]
        if qkv is not None:
            s = xn.sum(axis=(1, 2)); s = s + s[lanes ^ 32]
            mean = s / C
            v = ((xn - mean[:, None, None]) ** 2).sum(axis=(1, 2)); v = v + v[lanes ^ 32]
            rstd = 1.0 / np.sqrt(v / C + eps_next)
            for n in range(NTI):
                for be in range(2):
                    f = np.zeros((64, 8))
                    for s8 in range(8):
                        f[:, s8] = (xn[:, n, (2 * be + (s8 >> 2)) * 4 + (s8 & 3)] - mean) * rstd
                    xf[th][2 * n + be] = f
    if qkv is not None:
        heads, hd, Tp, qscale = qkv["heads"], qkv["hd"], qkv["Tp"], qkv["qscale"]
        for pi in range(3 * NTI):
            piece, sg, nl0 = Q0 + pi, pi // NTI, (pi % NTI) * 32
            for th in range(NH):
                if sg < 2:
                    acc = bias16(bqs[sg * C + nl0:sg * C + nl0 + 32])
                    for t in range(KS):
                        acc = mfma_32x32x16(stream[piece, t], xf[th][t], acc)
                    for lane in range(64):
                        j, h = lane & 31, lane >> 5
                        if 32 * th + j >= nvalid:
                            continue
                        tok = t0 + 32 * th + j
                        img, wi = divmod(tok, Tp)
                        for r in range(16):
                            f = nl0 + 8 * (r // 4) + 4 * h + (r % 4)
                            hh, dd = divmod(f, hd)
                            writes[("q" if sg == 0 else "k", ((img * heads + hh) * Tp + wi) * hd + dd)] = acc[lane, r] * (qscale if sg == 0 else 1.0)
                else:
                    acc = np.repeat(bqs[2 * C + nl0 + J][:, None], 16, axis=1)
                    for t in range(KS):
                        acc = mfma_32x32x16(xf[th][t], stream[piece, t], acc)
                    for lane in range(64):
                        j, h = lane & 31, lane >> 5
                        f = nl0 + j
                        hh, dd = divmod(f, hd)
                        for r in range(16):
                            tl = 32 * th + 8 * (r // 4) + 4 * h + (r % 4)
                            if tl >= nvalid:
                                continue
                            img, wv = divmod(t0 + tl, Tp)
                            writes[("v", ((img * heads + hh) * hd + dd) * Tp + wv)] = acc[lane, r]
    return out, writes
