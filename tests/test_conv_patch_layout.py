"""CPU: lane-level emulation of conv3x3_patch_kernel (lw-detr_amd/csrc/gemm.hip) against F.conv2d - the padded row image of the
patch, the tap's row offset into it, the per-lane validity masks, the (N, 9 Cin) weight order (tap-major, then channel: the order
`ConvX` weights are packed in, engine.py) and the accumulator layout of the 32x32x16 MFMA. float64: index arithmetic only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.vitblock_sim import mfma_32x32x16

LANES = np.arange(64)
L31, H = LANES & 31, LANES >> 5


@pytest.mark.parametrize("b,hp,wp,cin", [(2, 10, 13, 32), (1, 9, 16, 64), (3, 5, 7, 32)])
def test_patch_resident_conv_layout(b, hp, wp, cin):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(b, hp, wp, cin, generator=g, dtype=torch.float64)
    w = torch.randn(32, cin, 3, 3, generator=g, dtype=torch.float64)            # one 32-channel output tile is enough for the layout
    wk = w.permute(0, 2, 3, 1).reshape(32, -1).numpy()                            # (N, 9 Cin): k = tap * Cin + c
    rows = x.reshape(-1, cin).numpy()
    m_total, spr = rows.shape[0], cin // 8 + 1                                    # slots per padded patch row
    out = np.zeros((m_total, 32))
    for m0 in range(0, m_total, 128):                                             # one workgroup: 128 consecutive pixels (may straddle images)
        pr = 128 + 2 * wp + 2
        patch = np.zeros((pr, spr * 8))
        for r in range(pr):                                                       # the DMA: whole rows, zero outside [0, M)
            gm = m0 - wp - 1 + r
            if 0 <= gm < m_total:
                patch[r, :cin] = rows[gm]
        for tile in range(4):                                                     # 32-pixel tiles of the workgroup
            m = m0 + 32 * tile + L31
            r_img = m % (hp * wp); y, xx = r_img // wp, r_img % wp
            acc = np.zeros((64, 16))
            for tap in range(9):
                dy, dx = tap // 3 - 1, tap % 3 - 1
                top, bot, lef, rig = y == 0, y == hp - 1, xx == 0, xx == wp - 1
                row_ok = np.where(top, 0, 7) | 7 << 3 | np.where(bot, 0, 7) << 6
                col_ok = np.where(lef, 0, 0x49) | 0x92 | np.where(rig, 0, 0x124)
                valid = ((row_ok & col_ok) >> tap) & 1
                prow = 32 * tile + L31 + wp + 1 + dy * wp + dx                    # the tap is a row offset into the patch
                for c16 in range(cin // 16):
                    xf = np.stack([patch[prow[l], 16 * c16 + 8 * H[l]:16 * c16 + 8 * H[l] + 8] * valid[l] for l in range(64)])
                    wf = np.stack([wk[L31[l], tap * cin + 16 * c16 + 8 * H[l]:tap * cin + 16 * c16 + 8 * H[l] + 8] for l in range(64)])
                    acc = mfma_32x32x16(wf, xf, acc)                              # D[n][m]: register 4 q + r of lane (pixel, hi) is channel 8 q + 4 hi + r
            for l in range(64):
                if m[l] < m_total:
                    for e in range(16):
                        out[m[l], 8 * (e >> 2) + 4 * H[l] + (e & 3)] = acc[l, e]
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1).reshape(-1, 32).numpy()
    assert np.abs(out - ref).max() < 1e-9
