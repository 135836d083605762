"""CPU: the C-ABI library builds/loads and exports every symbol of include/lwdetr_hip.h; host-side logic
(state-dict compatibility, layout helpers, packed weights, post-processing) without touching a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import lwdetr_amd
from helpers import CASES, ROOT, golden_state_dict, load_golden


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    from lwdetr_amd import _native
    return _native


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "lwdetr_hip.h")).read()
    declared = set(re.findall(r"\b(lwdetr_[a-z0-9_]+)\s*\(", hdr))
    assert {"lwdetr_msda_forward", "lwdetr_gemm", "lwdetr_attention", "lwdetr_layernorm"} <= declared
    lib = ctypes.CDLL(built.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in include/lwdetr_hip.h but not exported"


def test_ctypes_structs_match_header_layout(built):
    """sizeof() of the ctypes mirrors equals what a C compiler computes for the header's structs."""
    import subprocess
    import tempfile
    src = '#include <stdio.h>\n#include "lwdetr_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(lwdetr_tok_layout),' \
          ' sizeof(lwdetr_gemm_seg), sizeof(lwdetr_gemm_desc), sizeof(lwdetr_attn_desc), sizeof(lwdetr_resize_image));return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(td, "s.c"), "-o", os.path.join(td, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(td, "s")]).split()]
    from lwdetr_amd.preprocess import ResizeImage
    assert sizes == [ctypes.sizeof(built.TokLayout), ctypes.sizeof(built.GemmSeg), ctypes.sizeof(built.GemmDesc),
                     ctypes.sizeof(built.AttnDesc), ctypes.sizeof(ResizeImage)]


def test_prof_api_without_gpu(built):
    lib = built.lib()
    n = lib.lwdetr_prof_num_kernels()
    names = [lib.lwdetr_prof_kernel_name(i).decode() for i in range(n)]
    assert "gemm_mfma" in names and "attn_global" in names and "msda_fused_forward" in names
    assert built.prof_collect() == {}


@pytest.mark.parametrize("name", ["tiny_640", "small_640", "medium_640", "large_640", "xlarge_640"])
def test_state_dict_keys_and_shapes_match_reference(name):
    g = load_golden(name)
    model, criterion, post = lwdetr_amd.build_model(lwdetr_amd.get_args(CASES[name][0]))
    sd = model.state_dict()
    assert sorted(sd) == [str(k) for k in g["sd_keys"]]
    ref_shapes = {str(k): str(s) for k, s in zip(g["sd_keys"], g["sd_shapes"])}
    assert all(",".join(map(str, v.shape)) == ref_shapes[k] for k, v in sd.items())
    model.load_state_dict(golden_state_dict(g), strict=True)
    assert set(post) == {"bbox"} and post["bbox"].num_select == lwdetr_amd.get_args(CASES[name][0]).num_select


def test_forward_fails_loudly_off_gpu():
    from lwdetr_amd._native import NativeError
    model, _, _ = lwdetr_amd.build_model(lwdetr_amd.get_args("tiny"))
    with pytest.raises(NativeError):
        model(torch.zeros(1, 3, 128, 128))


def test_window_major_position_embedding_matches_reference_order():
    """abs_pos_winmajor == reference get_abs_pos + reshape/permute (vit.py:26-54, 353-358), incl. pad rows."""
    from lwdetr_amd.engine import abs_pos_winmajor
    from oracle.lwdetr_torch import abs_pos
    c, hp, wp = 8, 12, 8
    pe = torch.randn(1, 197, c)
    h, w = hp // 4, wp // 4
    ref = abs_pos(pe, hp, wp).reshape(1, 4, h, 4, w, c).permute(0, 1, 3, 2, 4, 5).reshape(16, h * w, c)
    got = abs_pos_winmajor(pe, hp, wp, 8).reshape(16, 8, c)
    assert torch.allclose(got[:, :h * w], ref, atol=1e-6) and got[:, h * w:].abs().max() == 0


def test_convx_folding_and_gemm_weight_layouts():
    """BN folding + (Cout, tap*Cin) layout reproduce conv->BN; deconv layout reproduces conv_transpose2d."""
    from lwdetr_amd.engine import PackedWeights
    g = torch.Generator().manual_seed(0)
    sd = {"p.conv.weight": torch.randn(6, 4, 3, 3, generator=g), "p.bn.weight": torch.rand(6, generator=g) + 0.5,
          "p.bn.bias": torch.randn(6, generator=g), "p.bn.running_mean": torch.randn(6, generator=g),
          "p.bn.running_var": torch.rand(6, generator=g) + 0.5}
    pw = PackedWeights(sd, None, torch.device("cpu"), torch.float32)
    wk, b = pw.convx("p")
    x = torch.randn(2, 4, 5, 7, generator=g)
    ref = F.batch_norm(F.conv2d(x, sd["p.conv.weight"], padding=1), sd["p.bn.running_mean"], sd["p.bn.running_var"],
                       sd["p.bn.weight"], sd["p.bn.bias"], False, eps=1e-5)
    cols = F.unfold(x, 3, padding=1).view(2, 4, 9, 35).permute(0, 3, 2, 1).reshape(2, 35, 36)     # k = tap*Cin + ci
    got = (cols @ wk.t() + b).permute(0, 2, 1).reshape(2, 6, 5, 7)
    assert torch.allclose(got, ref, atol=1e-5)
    wd = torch.randn(4, 3, 2, 2, generator=g)
    wl = wd.permute(2, 3, 1, 0).reshape(12, 4)                          # engine layout: n = (dy*2+dx)*Cout + co
    y = (x.permute(0, 2, 3, 1) @ wl.t()).reshape(2, 5, 7, 2, 2, 3).permute(0, 5, 1, 3, 2, 4).reshape(2, 3, 10, 14)
    assert torch.allclose(y, F.conv_transpose2d(x, wd, stride=2), atol=1e-5)


def test_mlp_weight_packing_reproduces_the_unpacked_math():
    """Host packing for the block kernel / decoder FFN: LayerNorm affine folded into fc1, fc2 chunk-major in MFMA k-slot
    order; with ln_w = None (decoder FFN) fc1 is untouched. Any hidden size that is a multiple of 32 (not only 4 C)."""
    from lwdetr_amd import kernels as K
    g = torch.Generator().manual_seed(1)
    for c, hid, with_ln in [(64, 256, True), (64, 96, False), (32, 64, False)]:
        w1, b1 = torch.randn(hid, c, generator=g), torch.randn(hid, generator=g)
        w2 = torch.randn(c, hid, generator=g)
        lw, lb = (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)) if with_ln else (None, None)
        w1p, b1p, w2c = K.pack_mlp_weights(w1, b1, w2, lw, lb, torch.float32)
        x = torch.randn(5, c, generator=g)
        xin = F.layer_norm(x, (c,)) * lw + lb if with_ln else x
        xn = F.layer_norm(x, (c,)) if with_ln else x
        assert torch.allclose(xn @ w1p.t() + b1p, xin @ w1.t() + b1, atol=1e-4)
        assert w2c.shape == (hid // 32, c, 32)
        perm = torch.tensor(K._KSLOT_PERM)
        h = torch.randn(5, hid, generator=g)
        hp = h.view(5, hid // 32, 32)[:, :, perm]                        # the kernel holds a chunk's hidden units in k-slot order
        assert torch.allclose(torch.einsum("mkj,kcj->mc", hp, w2c), h @ w2.t(), atol=1e-4)
    assert K.ffn_fused_supported(256, 2048, torch.float16) and K.ffn_fused_supported(384, 2048, torch.bfloat16)
    assert not K.ffn_fused_supported(256, 2048, torch.float32) and not K.ffn_fused_supported(192, 768, torch.float16)


def test_postprocess_matches_golden_on_reference_outputs():
    g = load_golden("small_640")
    post = lwdetr_amd.models.PostProcess(300)
    out = {"pred_logits": torch.from_numpy(g["pred_logits"]), "pred_boxes": torch.from_numpy(g["pred_boxes"])}
    res = post(out, torch.tensor([[480.0, 640.0]] * 2))
    assert np.allclose(torch.stack([r["scores"] for r in res]).numpy(), g["post_scores"], atol=1e-6)
    assert np.array_equal(torch.stack([r["labels"] for r in res]).numpy(), g["post_labels"])
    assert np.allclose(torch.stack([r["boxes"] for r in res]).numpy(), g["post_boxes"], atol=1e-3)


def test_nested_tensor_padding_contract():
    from lwdetr_amd.models import nested_tensor_from_tensor_list
    nt = nested_tensor_from_tensor_list([torch.ones(3, 64, 128), torch.ones(3, 128, 64)])
    assert nt.tensors.shape == (2, 3, 128, 128) and nt.mask.dtype == torch.bool
    assert not nt.mask[0, :64, :].any() and nt.mask[0, 64:, :].all() and nt.mask[1, :, 64:].all()
    t, m = nested_tensor_from_tensor_list(torch.zeros(2, 3, 64, 64)).decompose()
    assert not m.any()


def test_synthetic_weights_are_machine_independent():
    """The integer-hash generator is pinned bit-for-bit (goldens were produced from exactly these tensors)."""
    import zlib
    from lwdetr_amd.synth import synth_images, synth_param, uniform01
    assert uniform01(12345, 3).tolist() == [0.01582909387275222, 0.5738614103347701, 0.10545253817227374]
    v = synth_param("backbone.0.encoder.blocks.0.attn.qkv.weight", (576, 192))
    assert zlib.crc32(v.numpy().tobytes()) == 2369530295
    assert zlib.crc32(synth_images(1, 64, 64).numpy().tobytes()) == 1422412767


def test_packed_weight_cache_token_sees_every_way_weights_can_change():
    """the token must move on sub-module conversions, `p.data = ...`, in-place updates of parameters AND
    buffers (BatchNorm statistics are folded into packed conv weights); and train() is checked on every forward."""
    import pytest as _pytest
    m, _, _ = lwdetr_amd.build_model(lwdetr_amd.get_args("tiny"))
    t0 = m._weights_token()
    assert m._weights_token() == t0
    m.backbone.half()
    t1 = m._weights_token()
    assert t1 != t0
    m.float()
    t2 = m._weights_token()
    m.class_embed.weight.data = m.class_embed.weight.data.clone()
    t3 = m._weights_token()
    assert t3 != t2
    bn = next(mod for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d))
    bn.running_var.mul_(1.5)
    assert m._weights_token() != t3
    m.train()
    with _pytest.raises(NotImplementedError):
        m._packed_weights()


def test_layernorm_fold_is_the_layernorm_then_linear():
    """kernels.fold_layernorm (host side of the opt-in LayerNorm fold, vit.py:199 / :217 in front of the QKV / fc1 Linear): with the weights it
    returns, rstd (x . w'_n - mean colsum_n) + b'_n - what lwdetr_gemm's ln_stats epilogue computes from the RAW rows - equals
    Linear(LayerNorm(x)) on those 16-bit weights (fp64; to the f32 rounding of the column sums), and differs from the unrounded f32 weights only by their 16-bit rounding."""
    from lwdetr_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    m, c, n, eps = 37, 256, 96, 1e-6
    x = (torch.randn(m, c, generator=g) * 2 + 0.5).half().double()
    w, b = torch.randn(n, c, generator=g) * c ** -0.5, torch.randn(n, generator=g) * 0.1
    ln_w, ln_b = torch.randn(c, generator=g) * 0.2 + 1, torch.randn(c, generator=g) * 0.1
    wq, colsum, bq = K.fold_layernorm(w, b, ln_w, ln_b, torch.float16)
    assert wq.dtype == torch.float16 and colsum.dtype == torch.float32 and bq.dtype == torch.float32 and wq.shape == (n, c)
    mean = x.mean(1, keepdim=True)
    rstd = (x.var(1, unbiased=False, keepdim=True) + eps).rsqrt()
    folded = rstd * (x @ wq.double().t() - mean * colsum.double()[None, :]) + bq.double()[None, :]
    # the same rounded weights applied the reference's way: LayerNorm without affine, then w' and b'
    xn = (x - mean) * rstd
    assert (folded - (xn @ wq.double().t() + bq.double())).abs().max().item() < 1e-6      # colsum is an f32 sum: ~1e-8 x |mean|
    # against the reference formulation with the f32 masters: only the 16-bit rounding of w' separates them
    ref = torch.nn.functional.layer_norm(x, (c,), ln_w.double(), ln_b.double(), eps) @ w.double().t() + b.double()
    assert (folded - ref).abs().max().item() < 2e-2 and ((folded - ref).abs().mean() / ref.abs().mean()).item() < 1e-3
    assert torch.equal(colsum, wq.float().sum(1))                 # column sums of the ROUNDED weights: what the MFMA contraction of a constant row yields
    wq0, _, bq0 = K.fold_layernorm(w, None, ln_w, ln_b, torch.bfloat16)
    assert wq0.dtype == torch.bfloat16 and torch.allclose(bq0, w @ ln_b, atol=1e-6)


def test_launch_chain_policy():
    """LWDETR._chains_for / set_streams: default two chains from 32 images (even batches only), 1 = always one, n = n chains
    whenever the parts have at least 8 images."""
    from lwdetr_amd.models import lwdetr as L
    from lwdetr_amd.models.lwdetr import LWDETR
    try:
        L.set_streams(0)
        assert [LWDETR._chains_for(b) for b in (1, 8, 16, 31, 32, 33, 64)] == [1, 1, 1, 1, 2, 1, 2]
        assert [LWDETR._chains_for(b, 960, 960) for b in (8, 14, 16, 32)] == [1, 1, 1, 2]         # the resolution does not change the rule
        L.set_streams(1)
        assert [LWDETR._chains_for(b) for b in (32, 64)] == [1, 1]
        L.set_streams(4)
        assert [LWDETR._chains_for(b) for b in (16, 32, 30, 64)] == [1, 4, 1, 4]
    finally:
        L.set_streams(0)


def test_conv_patch_kernel_index_arithmetic():
    """Host-side restatement of conv3x3_patch_kernel's integer arithmetic (lw-detr_amd/csrc/gemm.hip): the magic division it uses
    for pixel coordinates and patch slots, the tap-validity words built from four edge flags, and the linear-patch addressing
    (tap (dy, dx) of pixel m is patch row (m - m0) + W + 1 + dy W + dx) - against the plain definitions."""
    import numpy as np
    rng = np.random.default_rng(0)
    for dvs in (17, 25, 40 * 40, 80 * 80, 40, 80, 7, 2, 120 * 120):
        magic = (1 << 32) // dvs
        n = np.concatenate([rng.integers(0, 1 << 31, 20000, dtype=np.int64), np.arange(0, 4 * dvs), np.array([(1 << 31) - 1])])
        q = (n * magic) >> 32
        r = n - q * dvs
        q, r = np.where(r >= dvs, q + 1, q), np.where(r >= dvs, r - dvs, r)
        assert np.array_equal(q, n // dvs) and np.array_equal(r, n % dvs), dvs
    for h, w in ((40, 40), (10, 13), (7, 9), (2, 2), (1, 5)):
        for y in range(h):
            for x in range(w):
                top, bot, lef, rig = y == 0, y == h - 1, x == 0, x == w - 1
                row_ok = [0 if top else 7, 7, 0 if bot else 7]
                col_ok = (0 if lef else 0x49) | 0x92 | (0 if rig else 0x124)
                vbits = (row_ok[0] | row_ok[1] << 3 | row_ok[2] << 6) & col_ok
                for tap in range(9):
                    iy, ix = y + tap // 3 - 1, x + tap % 3 - 1
                    assert bool((vbits >> tap) & 1) == (0 <= iy < h and 0 <= ix < w), (h, w, y, x, tap)
    # patch rows: a tile of 128 consecutive pixels starting at m0 keeps global rows m0 - W - 1 .. m0 + 128 + W
    w_img, m0 = 13, 256
    for local in (0, 5, 127):
        for tap in range(9):
            pr = local + w_img + 1 + (tap // 3 - 1) * w_img + (tap % 3 - 1)
            assert 0 <= pr < 128 + 2 * w_img + 2
            assert m0 - w_img - 1 + pr == m0 + local + (tap // 3 - 1) * w_img + (tap % 3 - 1)
