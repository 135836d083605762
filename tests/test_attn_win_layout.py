"""CPU: lane-level emulation of attn_win_kernel's register layouts (lw-detr_amd/csrc/attention.hip) against dense softmax
attention - the score tile's register <-> key mapping, P packed in place as the B operand of the second MFMA, the key order of the
V^T operand that goes with it, the ones row that yields the denominator at hd 16, and the half-wave exchange in front of the
16-byte stores. Arithmetic is float64: this checks index arithmetic, not numerics (tests/test_gpu_kernels.py does those)."""
import numpy as np
import pytest

from tests.vitblock_sim import mfma_32x32x16

LANES = np.arange(64)
L31, H = LANES & 31, LANES >> 5


def frag_rows(mat, row0, nrows_valid, k0):
    """A / B operand of a 32x32x16 MFMA: lane (i = l & 31, h) holds mat[row0 + i][k0 + 8 h .. + 7] (rows clamped like the kernel)."""
    rows = np.minimum(row0 + L31, nrows_valid - 1)
    return np.stack([mat[rows[l], k0 + 8 * H[l]:k0 + 8 * H[l] + 8] for l in range(64)])


@pytest.mark.parametrize("hd,nkeys,nvalid", [(16, 100, 100), (16, 28, 25), (32, 100, 97), (32, 128, 126), (16, 8, 5)])
def test_one_wave_window_attention_layout(hd, nkeys, nvalid):
    rng = np.random.default_rng(1)
    q = rng.standard_normal((nkeys, hd)); k = rng.standard_normal((nkeys, hd)); v = rng.standard_normal((nkeys, hd))
    vt = v.T.copy()                                                    # (hd, keys), as the kernel reads it
    nc = hd // 16
    out = np.full((nkeys, hd), np.nan)
    nqt, nkt, nch = (nkeys + 31) // 32, (nvalid + 31) // 32, (nvalid + 15) // 16
    for qt in range(nqt):
        sc = []
        for kt in range(nkt):
            acc = np.zeros((64, 16))
            for c in range(nc):
                acc = mfma_32x32x16(frag_rows(k, 32 * kt, nkeys, 16 * c), frag_rows(q, 32 * qt, nkeys, 16 * c), acc)
            for e in range(16):                                        # register e of lane (query, h) is key 32 kt + 8 (e >> 2) + 4 h + (e & 3)
                key = 32 * kt + 8 * (e >> 2) + 4 * H + (e & 3)
                acc[:, e] = np.where(key < nvalid, acc[:, e], -np.inf)
            sc.append(acc)
        mx = np.max(np.stack(sc), axis=(0, 2)); mx = np.maximum(mx, mx[LANES ^ 32])
        o = np.zeros((64, 16)); lsum = np.zeros(64)
        for c in range(nch):
            pf = np.exp2(sc[c >> 1][:, 8 * (c & 1):8 * (c & 1) + 8] - mx[:, None])          # registers [8 (c & 1), + 8) of tile c >> 1, in place
            lsum += pf.sum(1)
            vf = np.zeros((64, 8))
            for l in range(64):
                row, h = L31[l], H[l]
                if hd == 16 and row >= 16:
                    vf[l] = 1.0 if row == 16 else 0.0
                else:                                                 # k-slot j is key 16 c + 8 (j >> 2) + 4 h + (j & 3): two runs of four
                    k0, k1 = 16 * c + 4 * h, 16 * c + 8 + 4 * h
                    k0 = k0 if k0 + 4 <= nkeys else nkeys - 4; k1 = k1 if k1 + 4 <= nkeys else nkeys - 4
                    vf[l] = np.concatenate([vt[row, k0:k0 + 4], vt[row, k1:k1 + 4]])
            o = mfma_32x32x16(vf, pf, o)
        if hd == 16:
            denom = np.where(H == 0, o[:, 8], o[LANES ^ 32, 8])         # channel 16 = register 8 of the lower half-wave
        else:
            denom = lsum + lsum[LANES ^ 32]
        y = o / denom[:, None]
        for bp in range(1 if hd == 16 else 2):                        # register 4 b + e of lane (query, h) is channel 8 b + 4 h + e
            lo, hi = y[:, 8 * bp:8 * bp + 4], y[:, 8 * bp + 4:8 * bp + 8]
            for l in range(64):
                qi, h = 32 * qt + L31[l], H[l]
                if qi >= nkeys:
                    continue
                p = l ^ 32
                st = np.concatenate([lo[l], lo[p]]) if h == 0 else np.concatenate([hi[p], hi[l]])   # after the half-wave exchange
                out[qi, 16 * bp + 8 * h:16 * bp + 8 * h + 8] = st
    s = (q @ k.T) * np.log(2.0)                                        # the kernel works in the log2 domain
    s[:, nvalid:] = -np.inf
    p = np.exp(s - s.max(1, keepdims=True)); ref = (p / p.sum(1, keepdims=True)) @ v
    assert np.abs(out - ref).max() < 1e-9
