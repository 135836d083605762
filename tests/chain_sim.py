"""Lane-level emulation of lw-detr_amd/csrc/chain.hip:enc_chain_kernel on the CPU (test infrastructure, numpy, float64).

Walks the PACKED weight stream exactly as the kernel does - global fragment g = 4 * piece + f is the 1 KB at g * 1024 bytes, lane l owns
elements [8 l, 8 l + 8) - with the gfx950 32x32x16 MFMA register layouts (tests/vitblock_sim.py:mfma_32x32x16), so the host-side packing
(lwdetr_amd.kernels.pack_enc_chain) and the kernel's index arithmetic are checked against the dense formulation without a GPU.
No 16-bit rounding: this validates layouts, not numerics."""
import numpy as np

from vitblock_sim import mfma_32x32x16


def simulate_enc_wave(stream, vec, rows_in, rowvalid, notpad, D, k5, nl, ncls, eps_p, eps_e):
    """One wave: rows_in (32, k5 or D) input rows; rowvalid / notpad (32,) flags. Returns dict of per-row outputs:
    memory (32, D) (k5 only), values (nl, 32, D), om (32, D), cls (32, 96), cls_max (32,)."""
    PF = k5 > 0
    KS, NTI = D // 16, D // 32
    KSI = k5 // 16 if PF else KS
    frags = np.asarray(stream, dtype=np.float64).reshape(-1, 64, 8)            # global fragment, lane, element
    vec = np.asarray(vec, dtype=np.float64)
    off = 0
    if PF:
        b2s, gps, bps = vec[0:D], vec[D:2 * D], vec[2 * D:3 * D]
        off = 3 * D
    bes, ges, bts = vec[off:off + D], vec[off + D:off + 2 * D], vec[off + 2 * D:off + 3 * D]
    bcs = vec[off + 3 * D:off + 3 * D + 96]
    bvs = vec[off + 3 * D + 96:off + 3 * D + 96 + 6 * D]
    lanes = np.arange(64); J, H = lanes & 31, lanes >> 5
    g = [0]                                                                    # fragment stream position

    def bias16(src):
        out = np.zeros((64, 16))
        for r in range(16):
            out[:, r] = src[8 * (r // 4) + 4 * H + (r % 4)]
        return out

    def tile(x, nf, acc):
        for f in range(nf):
            acc = mfma_32x32x16(frags[g[0] + f], x[f], acc)
        g[0] += nf
        return acc

    def chan(n, r):                 # channel of accumulator register r of tile n, per lane
        return 32 * n + 8 * (r // 4) + 4 * H + (r % 4)

    def layernorm_to_frags(acc_tiles, gam, bet, eps):
        x = np.stack(acc_tiles, 1)                                             # (64, NTI, 16)
        s = x.sum(axis=(1, 2)); s = s + s[lanes ^ 32]
        mean = s / D
        v = ((x - mean[:, None, None]) ** 2).sum(axis=(1, 2)); v = v + v[lanes ^ 32]
        rstd = 1.0 / np.sqrt(v / D + eps)
        rows = np.zeros((32, D))
        xo = [None] * KS
        for n in range(NTI):
            y = np.zeros((64, 16))
            for r in range(16):
                c = chan(n, r)
                y[:, r] = (x[:, n, r] - mean) * rstd * gam[c] + bet[c]
                rows[J, c] = y[:, r]
            xo[2 * n], xo[2 * n + 1] = y[:, 0:8].copy(), y[:, 8:16].copy()     # registers 0..7 / 8..15 = the two k-steps of the tile
        return rows, xo

    # input rows as B fragments in natural k order
    xin = [np.stack([rows_in[J[l], 16 * t + 8 * H[l]:16 * t + 8 * H[l] + 8] for l in range(64)]) for t in range(KSI)]
    out = {}
    if PF:
        accs = []
        for n in range(NTI):
            a = tile(xin, KSI, bias16(b2s[32 * n:32 * n + 32]))
            accs.append(a / (1.0 + np.exp(-a)))
        out["memory"], xf = layernorm_to_frags(accs, gps, bps, eps_p)
    else:
        xf = xin
    vals = np.zeros((nl, 32, D))
    for vt in range(nl * NTI):
        a = tile(xf, KS, bias16(bvs[32 * vt:32 * vt + 32]))
        li, n = divmod(vt, NTI)
        a = a * notpad[J][:, None]
        for r in range(16):
            vals[li, J, chan(n, r)] = a[:, r]
    out["values"] = vals
    xm = [f * rowvalid[J][:, None] for f in xf]
    accs = [tile(xm, KS, bias16(bes[32 * n:32 * n + 32])) for n in range(NTI)]
    out["om"], xo = layernorm_to_frags(accs, ges, bts, eps_e)
    cls = np.zeros((32, 96))
    for n in range(3):
        a = tile(xo, KS, bias16(bcs[32 * n:32 * n + 32]))
        for r in range(16):
            cls[J, chan(n, r)] = a[:, r]
    out["cls"] = cls
    out["cls_max"] = cls[:, :ncls].max(1)
    out["fragments_consumed"] = g[0]
    return out


def simulate_row_chain_split(stream, vec, stages, x_rows, res_rows, q_rows, D):
    """One 32-row workgroup of lw-detr_amd/csrc/chain.hip:mlp_chain_split_kernel (4 waves, wave w computes tiles w, w + 4, ... of every
    stage; at D = 384 a tile is two k-half steps and the stream is packed k-half-major inside a group of 4 tiles). stages: dicts with
    kind ("full" | "side"), n (side: columns), flags res / relu / ln (eps) / addq. Returns the list of per-stage outputs (32, N).
    float64, no rounding: validates RowChainOp.pack and the kernel's piece / fragment / hand-over arithmetic."""
    KS, NTI, PPT = D // 16, D // 32, D // 64
    HALVES = 2 if D == 384 else 1
    PPH, KSH = PPT // HALVES, KS // HALVES
    frags = np.asarray(stream, dtype=np.float64).reshape(-1, 64, 8)
    vec = np.asarray(vec, dtype=np.float64)
    lanes = np.arange(64); J, H = lanes & 31, lanes >> 5

    def chan(n, r):
        return 32 * n + 8 * (r // 4) + 4 * H + (r % 4)

    def bias16(src):
        out = np.zeros((64, 16))
        for r in range(16):
            out[:, r] = src[8 * (r // 4) + 4 * H + (r % 4)]
        return out

    def own_rows(mat, n):             # rows in accumulator layout of tile n
        out = np.zeros((64, 16))
        for r in range(16):
            out[:, r] = mat[J, chan(n, r)]
        return out

    # operand in "LDS": KS fragments, natural k order at the start
    act = [np.stack([x_rows[J[l], 16 * t + 8 * H[l]:16 * t + 8 * H[l] + 8] for l in range(64)]) for t in range(KS)]
    pc, off, outs = 0, 0, []

    def run_tile(xf, pc0, x0, acc):
        for f in range(KSH):
            acc = mfma_32x32x16(frags[4 * pc0 + f], xf[x0 + f], acc)
        return acc

    for st in stages:
        if st["kind"] == "side":
            nt = (st["n"] + 31) // 32
            bsrc = vec[off:off + 32 * nt]; off += 32 * nt
            res_out = np.zeros((32, 32 * nt))
            for g0 in range(0, nt, 4):
                ng = min(4, nt - g0)
                accs = {w: bias16(bsrc[32 * (g0 + w):32 * (g0 + w) + 32]) for w in range(ng)}
                for hf in range(HALVES):
                    for w in range(ng):
                        accs[w] = run_tile(act, pc + w * PPH, hf * KSH, accs[w])
                    pc += ng * PPH
                for w in range(ng):
                    for r in range(16):
                        res_out[J, chan(g0 + w, r)] = accs[w][:, r]
            outs.append(res_out[:, :st["n"]])
            continue
        bsrc = vec[off:off + D]; off += D
        if st.get("ln") is not None:
            gam, bet = vec[off:off + D], vec[off + D:off + 2 * D]; off += 2 * D
        tiles = {}
        for i in range(NTI // 4):
            accs = {}
            for w in range(4):
                n = w + 4 * i
                accs[w] = bias16(bsrc[32 * n:32 * n + 32]) + (own_rows(res_rows, n) if st.get("res") else 0.0)
            for hf in range(HALVES):
                for w in range(4):
                    accs[w] = run_tile(act, pc + w * PPH, hf * KSH, accs[w])
                pc += 4 * PPH
            for w in range(4):
                tiles[w + 4 * i] = np.maximum(accs[w], 0.0) if st.get("relu") else accs[w]
        y = np.zeros((32, D))
        for n in range(NTI):
            for r in range(16):
                y[J, chan(n, r)] = tiles[n][:, r]
        if st.get("ln") is not None:
            mu = y.mean(1, keepdims=True)
            y = (y - mu) / np.sqrt(((y - mu) ** 2).mean(1, keepdims=True) + st["ln"]) * gam + bet
        outs.append(y.copy())
        nxt = y + q_rows if st.get("addq") else y
        for n in range(NTI):                      # written back as the fragments (2 n, 2 n + 1) of tile n: registers 0..7 / 8..15
            t = own_rows(nxt, n)
            act[2 * n], act[2 * n + 1] = t[:, 0:8].copy(), t[:, 8:16].copy()
    return outs, pc
