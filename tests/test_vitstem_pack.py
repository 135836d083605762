"""CPU: host packing of the ViT stem kernel (lwdetr_amd.kernels.pack_vit_stem) against a lane-level emulation of the patch phase of
lw-detr_amd/csrc/vitblock.hip:vit_stem_kernel (stream order, fragment layout, k = (channel, patch row, pixel)) and the dense
PatchEmbed Conv2d of models/backbone/vit.py:353-358; the QKV pieces are those of the block kernel's packer (tests/test_vitblock_pack.py)."""
import numpy as np
import pytest
import torch

from tests.vitblock_sim import mfma_32x32x16


@pytest.mark.parametrize("c", [192, 384])
def test_pack_vit_stem_patch_phase_through_lane_emulation(c):
    from lwdetr_amd import kernels as K
    g = torch.Generator().manual_seed(c)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g, dtype=torch.float64) * sc
    wpe, bpe = r(c, 3, 16, 16, sc=768 ** -0.5), r(c, sc=0.1)
    wqkv, qb, vb, lw, lb = r(3 * c, c, sc=c ** -0.5), r(c, sc=0.1), r(c, sc=0.1), r(c, sc=0.2) + 1, r(c, sc=0.1)
    stream, vec = K.pack_vit_stem(wpe, bpe, wqkv, qb, vb, lw, lb, torch.float64)
    ks, nti = c // 16, c // 32
    assert stream.numel() == (24 + 3 * nti) * ks * 512 and vec.numel() * 4 % 4096 == 0
    assert np.abs(vec[:c].numpy() - bpe.float().numpy()).max() < 1e-7
    st = stream.numpy()
    img = r(3, 16, 16 * 32)                                   # 32 tokens side by side: token j = columns 16 j .. 16 j + 15
    acc = [np.zeros((64, 16)) for _ in range(nti)]
    for s in range(12):                                       # ring step: pieces 2 s, 2 s + 1 = patch rows (k-steps) 4 s .. 4 s + 3
        for fi in range(2 * ks):
            piece, f = 2 * s + fi // ks, fi % ks
            kk, n = fi // nti, fi % nti
            t = 4 * s + kk
            ch, py = t >> 4, t & 15
            a = st[(piece * ks + f) * 512:(piece * ks + f + 1) * 512].reshape(64, 8)
            b = np.zeros((64, 8))
            for lane in range(64):
                j, h = lane & 31, lane >> 5
                b[lane] = img[ch, py, 16 * j + 8 * h:16 * j + 8 * h + 8].numpy()          # the 16-byte load of vit_stem_kernel:load_px
            acc[n] = mfma_32x32x16(a, b, acc[n])
    patches = img.reshape(3, 16, 32, 16).permute(2, 0, 1, 3).reshape(32, 768)             # (token, (channel, row, pixel))
    ref = (patches @ wpe.reshape(c, 768).t()).numpy()                                        # (token, channel)
    for n in range(nti):
        for lane in range(64):
            j, h = lane & 31, lane >> 5
            for reg in range(16):
                ch = 32 * n + 8 * (reg // 4) + 4 * h + reg % 4
                assert abs(acc[n][lane, reg] - ref[j, ch]) < 5e-6                           # the packer keeps f32 master copies


def test_pack_vit_stem_qkv_pieces_are_the_block_kernels():
    from lwdetr_amd import kernels as K
    c = 192
    g = torch.Generator().manual_seed(3)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    wqkv, qb, vb, lw, lb = r(3 * c, c, sc=c ** -0.5), r(c, sc=0.1), r(c, sc=0.1), r(c, sc=0.2) + 1, r(c, sc=0.1)
    stream, vec = K.pack_vit_stem(r(c, 3, 16, 16), r(c), wqkv, qb, vb, lw, lb, torch.float16)
    one = torch.ones(c)
    blk, bvec = K.pack_vit_block(r(c, c), r(c), one, r(4 * c, c), r(4 * c), r(c, 4 * c), r(c), one, one, r(c), torch.float16, qkv=(wqkv, qb, vb, lw, lb))
    nq = 3 * (c // 32) * (c // 16) * 512
    assert torch.equal(stream[-nq:], blk[-nq:])                                              # same k-slot order, same LayerNorm folding
    assert torch.equal(vec[c:4 * c], bvec[10 * c:13 * c])
