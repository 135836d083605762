"""Thin Python handles over the C ABI: descriptors are built once (``GemmOp`` / ``AttnOp`` / ``LayerNormOp``) and
replayed with one ctypes call per launch, so a forward pass is a flat list of pre-built launches."""
import ctypes as C
import math

import os

import torch

from . import _native as _nat
from ._native import (A_CONV3x3, A_PATCH16, A_PLAIN, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SILU,  # noqa: F401
                      OUT_DECONV2x2, OUT_HEADS, OUT_HEADS_T, OUT_LINEAR, OUT_TOKMAP, AttnDesc, GemmDesc, GemmSeg,
                      TokLayout)

LOG2E = 1.4426950408889634


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _pad8(v, n, fill):
    """The kernels read bias / LayerScale vectors with 16-byte loads in runs of 8: make [0, ceil8(n)) readable."""
    if v is None:
        return None
    assert v.dtype == torch.float32
    need = (n + 7) // 8 * 8
    if v.numel() >= need and v.data_ptr() % 16 == 0:
        return v
    out = torch.full((need,), fill, dtype=torch.float32, device=v.device)
    out[:min(n, v.numel())] = v.reshape(-1)[:n]
    return out


def tok_layout(winmajor=False, hp=0, wp=0, twp=0) -> TokLayout:
    return TokLayout(1 if winmajor else 0, hp, wp, twp)


def seg(out, n_begin, n_end, *, mode=OUT_LINEAR, ldo=0, bias=None, act=ACT_NONE, scale=1.0, gamma=None, res=None,
        ldres=0, res_mod=0, out2=None, ld2=0, rowmask=None, rowmask_after=False, p0=0, p1=0, p2=0, in_tok=None, out_tok=None,
        out_batch_stride=0, out_row_offset=0, ln_stats=None, ln_colsum=None) -> GemmSeg:
    """One output column segment of a GEMM (see lwdetr_gemm_seg in include/lwdetr_hip.h). ln_stats (2, M) f32, PLANAR (mean row, then rstd row) + ln_colsum (n) f32:
    LayerNorm folded into the GEMM (fold_layernorm packs the weights; RowStatsOp produces the planar (2, M) statistics)."""
    s = GemmSeg()
    bias, gamma = _pad8(bias, n_end - n_begin, 0.0), _pad8(gamma, n_end - n_begin, 1.0)
    assert (ln_stats is None) == (ln_colsum is None)
    if ln_stats is not None:
        assert ln_stats.dtype == torch.float32 and ln_stats.is_contiguous() and ln_colsum.dtype == torch.float32
        ln_colsum = _pad8(ln_colsum, n_end - n_begin, 0.0)
        s.ln_stats, s.ln_colsum = _ptr(ln_stats), _ptr(ln_colsum)
    s._keep = (bias, gamma, ln_stats, ln_colsum)           # padded copies must outlive the launch
    s.out, s.out2, s.res = _ptr(out), _ptr(out2), _ptr(res)
    s.bias, s.gamma, s.rowmask = _ptr(bias), _ptr(gamma), _ptr(rowmask)
    assert rowmask is None or rowmask.dtype == torch.uint8
    s.scale, s.act, s.mode, s.n_begin, s.n_end = float(scale), act, mode, n_begin, n_end
    s.ldo, s.ld2, s.ldres, s.res_mod = ldo, ld2, ldres, res_mod
    s.p0, s.p1, s.p2 = p0, p1, p2
    s.rowmask_after = 1 if rowmask_after else 0
    s.in_tok = in_tok if in_tok is not None else tok_layout()
    s.out_tok = out_tok if out_tok is not None else tok_layout()
    s.out_batch_stride, s.out_row_offset = out_batch_stride, out_row_offset
    return s


SPLITK_SLAB = 64 * 68          # floats of one partial 64 x 64 tile in the kernel's stage layout (gemm.hip: gemm_dma_kernel SPLITK)


def splitk_for(M, N, K, dtype, has_a2=False):
    """Workgroups per 64 x 64 tile for a few-row GEMM with a long contraction (round 5; the single-image latency path: the 3x3 convolutions
    of the projector are 36 dependent k-stages on 50 tiles): 16-bit, up to 3200 rows, K >= 512, and only while tiles x splits stay within
    one round of the chip; at least 8 stages of 32 per slice - that was the policy that was measured; it lost (see below), so the automatic
    answer is 1 and LWDETR_GEMM_SPLITK=n forces n slices where legal (tests, tuning)."""
    env = os.environ.get("LWDETR_GEMM_SPLITK")
    if dtype not in (torch.float16, torch.bfloat16) or has_a2 or K % 64 != 0:
        return 1
    if env is not None:
        # forced (tests, tuning): only where the launch can take the 64 x 64 ring kernel's split form at all - few rows (the workspace is
        # tiles x n x 17 KB: at M = 51 200, N = 576 it would be 245 MB per op), a build that carries the kernel, enough k-stages per slice
        # at either stage depth - and a malformed value means "off", not an exception out of a plan build (advisor r5)
        try:
            n = int(env)
        except ValueError:
            return 1
        if not _nat.lib().lwdetr_has_experiments() or M > 3200:
            return 1
        return n if (2 <= n <= 16 and K // 64 >= n) else 1
    # Measured (tools/lat_bs1.py, single image, LW-DETR-small as one HIP graph; profiles/r5e_single_image_latency.txt): NOT a gain - the
    # publish / acquire of the slabs (two agent-scope fences + 17 KB per slice through L2) costs what the shorter k-chain saves: 3x3
    # convolutions (K = 1152, 4 slices) 19.2 -> 17.5 us, every K <= 768 GEMM 1.5-2.5 us SLOWER; p50 0.867 ms without, 0.880 with this policy,
    # 0.930 / 0.980 ms with 2 / 3 slices everywhere. Off unless LWDETR_GEMM_SPLITK=n asks for it.
    return 1


class GemmOp:
    """out = epilogue(A_view(M,K) @ W(N,K)^T). Keeps references to every tensor it points at."""

    def __init__(self, A, W, M, N, K, segs, *, lda=None, A2=None, a_mode=A_PLAIN, a_tok=None, conv_cin=0,
                 conv_stride=1, a_col0=0, conv_hout=0, conv_wout=0, img_h=0, img_w=0, keep=(), splitk=None):
        """splitk: None = automatic (few rows + long contraction: see splitk_for), 0 / 1 = off, n >= 2 = n workgroups per 64 x 64 tile."""
        N_ = N
        assert W.dtype == A.dtype and W.is_contiguous()
        d = GemmDesc()
        d.A, d.A2, d.W = _ptr(A), _ptr(A2), _ptr(W)
        d.M, d.N, d.K = M, N_, K
        d.lda = lda if lda is not None else K
        d.a_mode = a_mode
        d.a_tok = a_tok if a_tok is not None else tok_layout()
        d.conv_cin, d.conv_stride, d.a_col0 = conv_cin, conv_stride, a_col0
        d.conv_hout, d.conv_wout, d.img_h, d.img_w = conv_hout, conv_wout, img_h, img_w
        d.nseg = len(segs)
        for i, s in enumerate(segs):
            d.seg[i] = s
        self.desc, self.dtype = d, _nat.dtype_code(A.dtype)
        ws = None
        nsplit = splitk_for(M, N_, K, A.dtype, A2 is not None) if splitk is None else int(splitk)
        if nsplit >= 2:
            tiles = ((M + 63) // 64) * ((N_ + 63) // 64)
            ws = torch.zeros(tiles * nsplit * SPLITK_SLAB + tiles + 4, dtype=torch.float32, device=W.device)   # slabs, then the (zero) counters
            d.splitk_ws, d.splitk = _ptr(ws), nsplit
        self._keep = (A, A2, W, segs, ws) + tuple(keep)
        self._fn = _nat.lib().lwdetr_gemm
        self._ref = C.byref(d)
        # few rows (one or two images), one plain LINEAR segment: the few-row kernel on a fragment-major copy of W (made here, once per plan)
        sg0 = segs[0]
        if (type(self) is GemmOp and len(segs) == 1 and A2 is None and nsplit < 2 and sg0.mode == OUT_LINEAR and not sg0.rowmask and not sg0.ln_stats
                and sg0.res_mod == 0 and N_ % 16 == 0 and K % 32 == 0 and a_mode == A_PLAIN and d.lda % 8 == 0 and sg0.ldo % 4 == 0
                and gemm_few_supported(A.dtype, M, A_PLAIN, 0, K)):
            Wf = pack_frag16(W)
            d.W = _ptr(Wf)
            self._keep = self._keep + (Wf,)
            self._fn = _nat.lib().lwdetr_gemm_few

    def __call__(self, stream=None):
        rc = self._fn(self._ref, self.dtype, stream if stream is not None else _nat.stream_ptr())
        if rc:
            _nat.check(rc, f"gemm M={self.desc.M} N={self.desc.N} K={self.desc.K} a_mode={self.desc.a_mode}")


GEMM_FEW_MAX_ROWS = 3200        # one or two 640 x 640 images (see gemm_few_supported)


def gemm_few_supported(dtype, M, a_mode=A_PLAIN, conv_cin=0, K=0) -> bool:
    """The launch-plan choice for lwdetr_gemm_few (the few-row kernel on fragment-major weights): the 3x3 convolutions of the projector at one or two
    images (19.3 -> 10.8 us each at one image), and plain single-segment Linear launches of a few hundred rows (the decoder's ref_point_head at one
    image: 11.5 -> 6.9, 8.5 -> 6.4 us; at 1600 rows the 64 x 64 ring kernel ties or wins - profiles/r6c_*). LWDETR_GEMM_FEW=2 takes every plain
    launch it can (tuning), =0 keeps lwdetr_gemm everywhere (A/B runs)."""
    mode = os.environ.get("LWDETR_GEMM_FEW", "1")
    if mode == "0" or dtype not in (torch.float16, torch.bfloat16) or M > GEMM_FEW_MAX_ROWS:
        return False
    if a_mode == A_CONV3x3:
        return conv_cin in (128, 192)
    return a_mode == A_PLAIN and K >= 256 and (mode == "2" or M <= 640)


class GemmFewOp(GemmOp):
    """GemmOp on lwdetr_gemm_few: ``W`` is pack_frag16 of the (N, K) matrix GemmOp takes; one LINEAR segment."""

    def __init__(self, *a, **k):
        k["splitk"] = 0
        super().__init__(*a, **k)
        self._fn = _nat.lib().lwdetr_gemm_few


class AttnOp:
    def __init__(self, Q, K, VT, out, *, B, heads, hd, Tp, ldo, seqs_per_img, seq_tok_stride, keys_per_seq,
                 sub_stride, sub_len, kind, vt_slack=False):
        d = AttnDesc()
        d.Q, d.K, d.VT, d.out, d.ldo = _ptr(Q), _ptr(K), _ptr(VT), _ptr(out), ldo
        d.B, d.heads, d.hd, d.Tp = B, heads, hd, Tp
        d.seqs_per_img, d.seq_tok_stride, d.keys_per_seq = seqs_per_img, seq_tok_stride, keys_per_seq
        d.sub_stride, d.sub_len, d.kind = sub_stride, sub_len, kind
        d.vt_slack = 1 if vt_slack else 0
        self.desc, self.dtype = d, _nat.dtype_code(Q.dtype)
        self._keep = (Q, K, VT, out)
        self._fn = _nat.lib().lwdetr_attention
        self._ref = C.byref(d)

    def __call__(self, stream=None):
        rc = self._fn(self._ref, self.dtype, stream if stream is not None else _nat.stream_ptr())
        if rc:
            _nat.check(rc, "attention")


class LayerNormOp:
    def __init__(self, x, gamma, beta, out, M, C_, eps, *, ldx=None, ldo=None, rows_per_batch=0, out_batch_rows=0,
                 out_row_offset=0):
        assert gamma.dtype == torch.float32 and beta.dtype == torch.float32
        self.args = (_ptr(x), ldx if ldx is not None else C_, _ptr(gamma), _ptr(beta), _ptr(out),
                     ldo if ldo is not None else C_, M, C_, float(eps), rows_per_batch, out_batch_rows,
                     out_row_offset, _nat.dtype_code(x.dtype))
        self._keep = (x, gamma, beta, out)
        self._fn = _nat.lib().lwdetr_layernorm

    def __call__(self, stream=None):
        rc = self._fn(*self.args, stream if stream is not None else _nat.stream_ptr())
        if rc:
            _nat.check(rc, "layernorm")


class RowStatsOp:
    """stats (2, M) f32, planar: stats[0, m] = mean, stats[1, m] = rstd of row m of x (M, C): the statistics half of a LayerNorm, for a GEMM
    with the LayerNorm folded in."""

    def __init__(self, x, stats, M, C_, eps, *, ldx=None):
        assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.numel() >= 2 * M
        self.args = (_ptr(x), ldx if ldx is not None else C_, M, C_, float(eps), _ptr(stats), _nat.dtype_code(x.dtype))
        self._keep = (x, stats)
        self._fn = _nat.lib().lwdetr_row_stats

    def __call__(self, stream=None):
        rc = self._fn(*self.args, stream if stream is not None else _nat.stream_ptr())
        if rc:
            _nat.check(rc, "row_stats")


def fold_layernorm(w, b, ln_w, ln_b, dtype):
    """LayerNorm (affine ln_w, ln_b) in front of Linear(w, b), folded for lwdetr_gemm's ln_stats epilogue:
    LN(x) w^T + b = rstd (x . w'_n - mean colsum_n) + b'_n with w' = w diag(ln_w) rounded to the compute dtype, colsum_n = sum_k w'_nk of the
    ROUNDED weights in f32 (what the MFMA contraction of a constant row yields), b' = b + w ln_b in f32. Returns (w', colsum, b')."""
    wf = w.detach().float()
    wq = (wf * ln_w.detach().float()[None, :]).to(dtype).contiguous()
    colsum = wq.float().sum(1).contiguous()
    bq = (wf @ ln_b.detach().float() + (b.detach().float() if b is not None else 0.0)).contiguous()
    return wq, colsum, bq


class LayerNormChainOp:
    """out1 = LN1(x); out2 = LN2(out1): one launch (lwdetr_layernorm_chain)."""

    def __init__(self, x, g1, b1, eps1, out1, g2, b2, eps2, out2, M, C_):
        assert all(t.dtype == torch.float32 for t in (g1, b1, g2, b2))
        self.args = (_ptr(x), C_, _ptr(g1), _ptr(b1), float(eps1), _ptr(out1), C_, _ptr(g2), _ptr(b2), float(eps2),
                     _ptr(out2), C_, M, C_, _nat.dtype_code(x.dtype))
        self._keep = (x, g1, b1, out1, g2, b2, out2)
        self._fn = _nat.lib().lwdetr_layernorm_chain

    def __call__(self, stream=None):
        rc = self._fn(*self.args, stream if stream is not None else _nat.stream_ptr())
        if rc:
            _nat.check(rc, "layernorm_chain")


def ffn_fused_supported(C_, hid, dtype) -> bool:
    """Shapes lwdetr_ffn_partial is instantiated for (the decoder widths of the shipped configurations)."""
    return C_ in (256, 384) and hid % 64 == 0 and dtype in (torch.float16, torch.bfloat16)


def ffn_partial_floats(M, C_, hid, dtype) -> int:
    """Floats of scratch one FfnOp of this shape needs (splits * M * C)."""
    splits = _nat.lib().lwdetr_ffn_splits(M, C_, hid, _nat.dtype_code(dtype))
    return max(splits, 0) * M * C_


class FfnOp:
    """Decoder FFN + norm3 (+ decoder.norm): x_out = LN1(x + W2 ReLU(W1 x + b1) + b2), hs = LN2(x_out); two launches
    (lwdetr_ffn_partial, lwdetr_ffn_finish). ``w1, b1, w2c`` from ``pack_mlp_weights(..., ln_w=None, ln_b=None)``."""

    def __init__(self, x, w1, b1, w2c, b2, g1, be1, eps1, out1, g2, be2, eps2, out2, M, C_, partial=None):
        hid = w1.shape[0]
        assert all(t.dtype == torch.float32 for t in (b1, b2, g1, be1)) and w1.dtype == x.dtype == w2c.dtype
        assert g2 is None or (g2.dtype == torch.float32 and be2.dtype == torch.float32)
        assert w1.is_contiguous() and w2c.is_contiguous()
        code = _nat.dtype_code(x.dtype)
        splits = _nat.lib().lwdetr_ffn_splits(M, C_, hid, code)
        if splits <= 0:
            _nat.check(-splits if splits < 0 else 2, "ffn_splits")
        self.splits = splits
        # the f32 partial slabs (splits x M x C): the decoder layers run back to back on one stream, so a launch plan hands
        # every layer the same scratch (``partial``, at least splits * M * C floats) instead of one allocation per layer
        if partial is None:
            partial = torch.empty(splits * M * C_, dtype=torch.float32, device=x.device)
        assert partial.dtype == torch.float32 and partial.numel() >= splits * M * C_ and partial.is_contiguous()
        self.partial = partial
        self.a_part = (_ptr(x), C_, _ptr(w1), _ptr(b1), _ptr(w2c), _ptr(self.partial), M, C_, hid, code)
        self.a_fin = (_ptr(x), C_, _ptr(self.partial), splits, _ptr(b2), _ptr(g1), _ptr(be1), float(eps1), _ptr(out1), C_,
                      _ptr(g2) if g2 is not None else None, _ptr(be2) if g2 is not None else None, float(eps2),
                      _ptr(out2) if g2 is not None else None, C_, M, C_, code)
        self._keep = (x, w1, b1, w2c, b2, g1, be1, out1, g2, be2, out2)
        self._f_part, self._f_fin = _nat.lib().lwdetr_ffn_partial, _nat.lib().lwdetr_ffn_finish

    def __call__(self, stream=None):
        st = stream if stream is not None else _nat.stream_ptr()
        rc = self._f_part(*self.a_part, st)
        if rc:
            _nat.check(rc, "ffn_partial")
        rc = self._f_fin(*self.a_fin, st)
        if rc:
            _nat.check(rc, "ffn_finish")


MLP_FUSED_MIN_ROWS = 12800      # below ~8 images of 640x640 the per-tile latency of the fused kernel loses to small GEMMs


def mlp_fused_supported(C_, dtype, rows=None) -> bool:
    """Shapes the fused MLP kernel is instantiated for (LDS / register budget): C=192 any dtype, C=384 16-bit.

    With ``rows`` given this is the launch-plan choice: the fused kernel gives every wave ONE 32-token tile and walks all
    hidden chunks with it, so its run time is flat (~70-100 us) up to 65k tokens; for a few images the 6 separate launches
    (LN, QKV, proj, LN, fc1+GELU, fc2) spread the same work over all CUs and win (bs=1 latency). LWDETR_MLP_FUSED=0/1
    forces either. (16-bit C = 192 is always fused: for a few images the C entry point uses a second kernel that makes one
    32-token tile a workgroup and splits the hidden dimension over its waves.)"""
    ok = C_ == 192 or (C_ == 384 and dtype in (torch.float16, torch.bfloat16))
    if not ok or rows is None:
        return ok
    force = os.environ.get("LWDETR_MLP_FUSED")
    if force in ("0", "1"):
        return force == "1"
    if C_ == 192 and dtype in (torch.float16, torch.bfloat16):
        return True             # lwdetr_mlp_fused switches to its tile-per-workgroup kernel below MLP_FUSED_MIN_ROWS by itself
    return rows >= MLP_FUSED_MIN_ROWS


_KSLOT_PERM = [4 * g_ + e + 16 * hi for g_ in range(4) for hi in range(2) for e in range(4)]


def pack_mlp_weights(w1, b1, w2, ln_w, ln_b, dtype, proj=False):
    """Host-side packing for lwdetr_mlp_fused (f32 master tensors in, device tensors of ``dtype`` / f32 out):
    LayerNorm's affine is folded into fc1, fc2 is re-laid out chunk-major (32 hidden units per contiguous tile)."""
    w1, b1, w2 = (t.float() for t in (w1, b1, w2))
    hid, c = w1.shape
    if ln_w is None:            # decoder FFN (lwdetr_ffn_partial): no LayerNorm in front of linear1
        b1f, w1f = b1.contiguous(), w1
    else:
        ln_w, ln_b = ln_w.float(), ln_b.float()
        b1f = (b1 + w1 @ ln_b).contiguous()
        w1f = w1 * ln_w[None, :]
    if proj:   # the fused projection hands x over in accumulator order: permute fc1's columns inside every 32-chunk
        w1f = w1f.view(hid, c // 32, 32)[:, :, torch.tensor(_KSLOT_PERM)].reshape(hid, c)
    w1f = w1f.to(dtype).contiguous()
    # chunk-major, and inside a chunk the MFMA k-slot order of the fused kernel: lane group g holds hidden
    # (4g..4g+3, 16+4g..16+4g+3) as one contiguous run of 8
    perm = torch.tensor(_KSLOT_PERM)
    w2c = w2.view(c, hid // 32, 32)[:, :, perm].permute(1, 0, 2).to(dtype).contiguous()
    return w1f, b1f, w2c


def pack_qkv_weights(wqkv, q_bias, v_bias, ln_w, ln_b, dtype):
    """Next-block QKV for the chained form of lwdetr_mlp_fused: LayerNorm affine folded in, columns in k-slot order."""
    wqkv, q_bias, v_bias, ln_w, ln_b = (t.float() for t in (wqkv, q_bias, v_bias, ln_w, ln_b))
    c = wqkv.shape[1]
    bias = torch.cat([q_bias, torch.zeros_like(q_bias), v_bias]) + wqkv @ ln_b
    w = (wqkv * ln_w[None, :]).view(3 * c, c // 32, 32)[:, :, torch.tensor(_KSLOT_PERM)].reshape(3 * c, c)
    return w.to(dtype).contiguous(), bias.contiguous()


def pack_frag16(w):
    """(R, K) row-major -> fragment-major [R / 16][K / 32][16][32]: every 16 x 32 MFMA A fragment one contiguous KB (few-token block kernel)."""
    r, k = w.shape
    return w.view(r // 16, 16, k // 32, 32).permute(0, 2, 1, 3).contiguous()


class MlpFusedOp:
    """x <- x + gamma2 * fc2(GELU(fc1(LN(x)))) in one launch (weights packed by ``pack_mlp_weights``)."""

    def __init__(self, x, w1f, b1f, w2c, b2, gamma2, M, C_, eps, *, ldx=None, out2=None, ld2=0, stats_out=None,
                 eps_next=1e-6, att=None, ldatt=None, wp=None, bp=None, gamma1=None, wqkv=None, bqkv=None, q=None,
                 k=None, vt=None, qscale=1.0, heads=0, hd=0, Tp=0):
        assert b1f.dtype == torch.float32 and b2.dtype == torch.float32 and gamma2.dtype == torch.float32
        assert w1f.dtype == x.dtype and w2c.dtype == x.dtype and w1f.is_contiguous() and w2c.is_contiguous()
        self.args = (_ptr(x), ldx if ldx is not None else C_, _ptr(w1f), _ptr(b1f), _ptr(w2c), _ptr(b2), _ptr(gamma2),
                     _ptr(out2), ld2, _ptr(stats_out), M, C_, float(eps), float(eps_next), _ptr(att),
                     ldatt if ldatt is not None else C_, _ptr(wp), _ptr(bp), _ptr(gamma1), _ptr(wqkv), _ptr(bqkv),
                     _ptr(q), _ptr(k), _ptr(vt), float(qscale), heads, hd, Tp, _nat.dtype_code(x.dtype))
        if att is not None:
            assert wp.dtype == x.dtype and wp.is_contiguous() and bp.dtype == torch.float32 and gamma1.dtype == torch.float32
        self._keep = (x, w1f, b1f, w2c, b2, gamma2, out2, stats_out, att, wp, bp, gamma1, wqkv, bqkv, q, k, vt)
        self._fn = _nat.lib().lwdetr_mlp_fused

    def __call__(self, stream=None):
        rc = self._fn(*self.args, stream if stream is not None else _nat.stream_ptr())
        if rc:
            _nat.check(rc, "mlp_fused")


VIT_BLOCK_FEW_MAX_ROWS = 12800      # = MLP_SMALL_MAX_ROWS of csrc/mlp.hip


def vit_block_few_supported(C_, dtype, rows) -> bool:
    """The launch-plan choice for lwdetr_vit_block_few (the few-token block kernel on fragment-major weights): 16-bit C = 192 below
    ~8 images of 640 x 640. LWDETR_VIT_BLOCK_FEW=0 keeps the row-major form (lwdetr_mlp_fused; A/B runs)."""
    return (C_ == 192 and dtype in (torch.float16, torch.bfloat16) and rows is not None and rows < VIT_BLOCK_FEW_MAX_ROWS
            and os.environ.get("LWDETR_VIT_BLOCK_FEW", "1") != "0")


class VitBlockFewOp(MlpFusedOp):
    """MlpFusedOp with the attention projection on fragment-major Wp / W1 / Wqkv (``pack_frag16`` of what MlpFusedOp takes): lwdetr_vit_block_few."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        assert k.get("att") is not None
        self._fn = _nat.lib().lwdetr_vit_block_few


# ---------------------------------------------------------------------------------------- fused ViT block (round 3)
VIT_BLOCK_MIN_ROWS = 12800      # below ~8 images of 640x640 the few-token kernels of lwdetr_mlp_fused win (bs=1 latency path)
_VB_MIN_GAMMA = 1e-12           # the kernel divides by gamma_1 / gamma_2 (f32): refuse zero / denormal LayerScale entries


def vit_block_supported(C_, dtype, hd=None, rows=None) -> bool:
    """Shapes / dtypes lwdetr_vit_block is instantiated for; with ``rows`` the launch-plan choice (LWDETR_VIT_BLOCK=0/1 forces)."""
    ok = C_ in (192, 384) and dtype in (torch.float16, torch.bfloat16) and (hd is None or (hd >= 8 and hd & (hd - 1) == 0))
    if not ok or rows is None:
        return ok
    force = os.environ.get("LWDETR_VIT_BLOCK")
    if force in ("0", "1"):
        return force == "1"
    return rows >= VIT_BLOCK_MIN_ROWS and rows % 8 == 0


def vb_kslot_channels(c):
    """Channel held by k-slot position p = 16 t + 8 h + s of an accumulator tile handed on as a B operand (vitblock.hip):
    t = 2 n + beta, s = 4 b' + e  <->  channel 32 n + 16 beta + 8 b' + 4 h + e."""
    idx = []
    for t in range(c // 16):
        n, be = divmod(t, 2)
        for h in range(2):
            for s_ in range(8):
                bp, e = divmod(s_, 4)
                idx.append(32 * n + 16 * be + 8 * bp + 4 * h + e)
    return torch.tensor(idx)


def _vb_frags(w32):
    """(32, K) rows in k-slot order -> (K/16, 2, 32, 8): fragment t, lane half h, row i, element s (= lane-linear 1 KB pieces)."""
    return w32.reshape(32, -1, 2, 8).permute(1, 2, 0, 3).contiguous()


def pack_vit_block(wp, bp, g1, w1, b1, w2, b2, g2, ln2_w, ln2_b, dtype, qkv=None, order=None):
    """Host-side packing for lwdetr_vit_block (f32 master tensors in): returns (stream of ``dtype``, vec f32).

    stream = pieces of 32 rows x C (or C x 32) in consumption order, every piece KS = C/16 fragments of 1 KB in MFMA lane
    order: Wp tiles 0..C/32-1 (natural k order: the attention rows come straight from memory); then the hidden pieces in the
    order of the kernel form - "pipelined" (C = 384, 4-wave kernel): W1c(0), W1c(1), (W2c(k-1), W1c(k+1)) for k = 1..NCH-2,
    W2c(NCH-2), W2c(NCH-1); "alternating" (C = 192, 8-wave kernel): pairs (W1c(0), pad), (W2c(k), W1c(k+1)) for k = 0..NCH-2, (W2c(NCH-1), pad)
    (W1 = fc1 * ln2_w with its columns in k-slot order,
    W2c(k) = fc2[:, 32k..32k+31] as fragments (k-step, output tile) with the hidden units in accumulator order); optionally
    the next block's Wqkv' tiles (``qkv`` = (wqkv, q_bias, v_bias, ln1_w, ln1_b), LayerNorm affine folded, k-slot order).
    vec = b1' | bp | g1 | 1/g1 | b2 | 1/g2 | g2 | bqkv' zero-padded to a multiple of 4 KB."""
    f = lambda t: t.detach().float().cpu()
    wp, bp, g1, w1, b1, w2, b2, g2, ln2_w, ln2_b = map(f, (wp, bp, g1, w1, b1, w2, b2, g2, ln2_w, ln2_b))
    c = wp.shape[0]
    hid = w1.shape[0]
    assert c in (192, 384) and hid == 4 * c and w2.shape == (c, hid)
    if min(g1.abs().min().item(), g2.abs().min().item()) < _VB_MIN_GAMMA:
        raise ValueError("lwdetr_vit_block needs non-zero LayerScale (gamma_1 / gamma_2)")
    nti, nch = c // 32, hid // 32
    perm = vb_kslot_channels(c)
    w1f = (w1 * ln2_w[None, :])[:, perm]
    b1f = b1 + w1 @ ln2_b
    pieces = [_vb_frags(wp[32 * n:32 * n + 32]) for n in range(nti)]
    w1c = lambda k: _vb_frags(w1f[32 * k:32 * k + 32])

    def w2c(k):
        # A2[n, i, kap, h, s] = w2[32 n + i, 32 k + 16 kap + 8 (s // 4) + 4 h + s % 4]; fragments ordered (kap, n), each (h, i, s)
        blk = w2[:, 32 * k:32 * k + 32].reshape(nti, 32, 2, 2, 2, 4)           # n, i, kap, b', h, e
        return blk.permute(2, 0, 4, 1, 3, 5).reshape(2 * nti, 2, 32, 8).contiguous()   # (kap, n), h, i, (b', e)

    order = order or "pipelined"        # "alternating": the 8-wave experiment of round 3 (profiles/r3b_*), not in the product
    if order == "pipelined":
        pieces += [w1c(0), w1c(1)]
        for k in range(1, nch - 1):
            pieces += [w2c(k - 1), w1c(k + 1)]
        pieces += [w2c(nch - 2), w2c(nch - 1)]
    else:
        assert order == "alternating"      # pairs of pieces, one pair per MFMA task: two zero pad pieces
        pad = torch.zeros_like(w1c(0))
        pieces += [w1c(0), pad]
        for k in range(nch - 1):
            pieces += [w2c(k), w1c(k + 1)]
        pieces += [w2c(nch - 1), pad]
    bq = torch.zeros(3 * c)
    if qkv is not None:
        wqkv, q_bias, v_bias, ln1_w, ln1_b = map(f, qkv)
        bq = torch.cat([q_bias, torch.zeros_like(q_bias), v_bias]) + wqkv @ ln1_b
        wq = (wqkv * ln1_w[None, :])[:, perm]
        pieces += [_vb_frags(wq[32 * i:32 * i + 32]) for i in range(3 * nti)]
    stream = torch.cat([p_.reshape(-1) for p_ in pieces]).to(dtype).contiguous()
    vec = torch.cat([b1f, bp, g1, 1.0 / g1, b2, 1.0 / g2, g2, bq])
    nvec = (13 * c * 4 + 4095) // 4096 * 4096 // 4
    vec = torch.cat([vec, torch.zeros(nvec - vec.numel())]).contiguous()
    return stream, vec


def pack_vit_qkv(wqkv, q_bias, v_bias, ln_w, ln_b, dtype):
    """Host-side packing for lwdetr_vit_qkv (norm1 + QKV of a block on its own: block 0): stream = 3 C / 32 pieces of 32 output
    features x C in NATURAL k order (the rows come straight from memory), LayerNorm affine folded; vec = [q_bias, 0, v_bias] + W beta."""
    f = lambda t: t.detach().float().cpu()
    wqkv, q_bias, v_bias, ln_w, ln_b = map(f, (wqkv, q_bias, v_bias, ln_w, ln_b))
    c = wqkv.shape[1]
    assert c in (192, 384) and wqkv.shape[0] == 3 * c
    bq = torch.cat([q_bias, torch.zeros_like(q_bias), v_bias]) + wqkv @ ln_b
    wq = wqkv * ln_w[None, :]
    stream = torch.cat([_vb_frags(wq[32 * i:32 * i + 32]).reshape(-1) for i in range(3 * c // 32)]).to(dtype).contiguous()
    nvec = (3 * c * 4 + 4095) // 4096 * 4096 // 4
    return stream, torch.cat([bq, torch.zeros(nvec - bq.numel())]).contiguous()


class VitQkvOp:
    """q, k, v^T = heads(LN(x) Wqkv^T + b): norm1 + QKV of a ViT block in one launch (lwdetr_vit_qkv)."""

    def __init__(self, x, stream, vec, M, C_, eps, *, q, k, vt, qscale, heads, hd, Tp, ldx=None):
        assert stream.dtype == x.dtype and vec.dtype == torch.float32
        lib = _nat.lib()
        assert stream.numel() * 2 == lib.lwdetr_vit_qkv_stream_bytes(C_) and vec.numel() == lib.lwdetr_vit_qkv_vec_floats(C_)
        self.args = (_ptr(x), ldx if ldx is not None else C_, _ptr(stream), _ptr(vec), M, C_, float(eps), _ptr(q), _ptr(k), _ptr(vt),
                     float(qscale), heads, hd, Tp, _nat.dtype_code(x.dtype))
        self._keep = (x, stream, vec, q, k, vt)
        self._fn = lib.lwdetr_vit_qkv

    def __call__(self, stream=None):
        rc = self._fn(*self.args, stream if stream is not None else _nat.stream_ptr())
        if rc:
            _nat.check(rc, "vit_qkv")


def pack_vit_stem(wpe, bpe, wqkv, q_bias, v_bias, ln_w, ln_b, dtype):
    """Host-side packing for lwdetr_vit_stem (patch embedding + norm1 + QKV of block 0): stream = 24 patch pieces - piece i holds the
    fragments (k-step 2 i + kk, channel tile n), kk-major, of Wpe (C, 768) with k = (channel, patch row, pixel) as the Conv2d weight
    lies in memory - then the 3 C / 32 QKV pieces in accumulator k-slot order (as pack_vit_block's chained QKV, LayerNorm affine
    folded); vec = patch bias | [q_bias, 0, v_bias] + W beta, zero-padded to a multiple of 4 KB."""
    f = lambda t: t.detach().float().cpu()
    wpe, bpe, wqkv, q_bias, v_bias, ln_w, ln_b = map(f, (wpe, bpe, wqkv, q_bias, v_bias, ln_w, ln_b))
    c = wpe.shape[0]
    wpe = wpe.reshape(c, -1)
    assert c in (192, 384) and wpe.shape[1] == 768 and wqkv.shape == (3 * c, c)
    nti = c // 32
    pieces = []
    for i in range(24):
        for kk in range(2):
            t = 2 * i + kk
            pieces += [_vb_frags(wpe[32 * n:32 * n + 32, 16 * t:16 * t + 16]) for n in range(nti)]
    perm = vb_kslot_channels(c)
    bq = torch.cat([q_bias, torch.zeros_like(q_bias), v_bias]) + wqkv @ ln_b
    wq = (wqkv * ln_w[None, :])[:, perm]
    pieces += [_vb_frags(wq[32 * i:32 * i + 32]) for i in range(3 * nti)]
    stream = torch.cat([p_.reshape(-1) for p_ in pieces]).to(dtype).contiguous()
    vec = torch.cat([bpe, bq])
    nvec = (4 * c * 4 + 4095) // 4096 * 4096 // 4
    return stream, torch.cat([vec, torch.zeros(nvec - vec.numel())]).contiguous()


class VitStemOp:
    """x0 = patches Wpe^T + b + pos (stored), q, k, v^T = heads(LN(x0) Wqkv^T + b): the ViT stem in one launch (lwdetr_vit_stem).
    ``img`` (B, 3, 16 Hp, 16 Wp) of the model dtype - ``set_image`` rebinds the pointer per call; ``pos`` (16 Twp, C) window-major."""

    def __init__(self, img, pos, x, stream, vec, B, Hp, Wp, Twp, C_, eps, *, q, k, vt, qscale, heads, hd, ldx=None):
        assert stream.dtype == x.dtype == pos.dtype and vec.dtype == torch.float32 and pos.is_contiguous() and pos.shape == (16 * Twp, C_)
        lib = _nat.lib()
        assert stream.numel() * 2 == lib.lwdetr_vit_stem_stream_bytes(C_) and vec.numel() == lib.lwdetr_vit_stem_vec_floats(C_)
        self._img = _ptr(img) if img is not None else None
        self.args = [None, B, 16 * Hp, 16 * Wp, Hp, Wp, Twp, _ptr(pos), C_, _ptr(x), ldx if ldx is not None else C_, _ptr(stream), _ptr(vec),
                     B * 16 * Twp, C_, float(eps), _ptr(q), _ptr(k), _ptr(vt), float(qscale), heads, hd, _nat.dtype_code(x.dtype)]
        self._keep = (img, pos, x, stream, vec, q, k, vt)
        self._fn = lib.lwdetr_vit_stem

    def set_image(self, ptr):
        self._img = ptr

    def __call__(self, stream=None):
        self.args[0] = self._img
        rc = self._fn(*self.args, stream if stream is not None else _nat.stream_ptr())
        if rc:
            _nat.check(rc, "vit_stem")


class VitBlockOp:
    """x <- block tail (attention projection + MLP, + norm1 / QKV of the next block) in one launch (lwdetr_vit_block)."""

    def __init__(self, x, att, stream, vec, M, C_, eps, *, ldx=None, ldatt=None, out2=None, ld2=0, stats_out=None, eps_next=1e-6,
                 q=None, k=None, vt=None, qscale=1.0, heads=0, hd=0, Tp=0):
        assert stream.dtype == x.dtype == att.dtype and vec.dtype == torch.float32 and stream.is_contiguous() and vec.is_contiguous()
        has_qkv = q is not None
        nti = C_ // 32
        assert stream.numel() == (nti + 8 * nti + (3 * nti if has_qkv else 0)) * (C_ // 16) * 512, "stream / qkv mismatch"
        self.args = (_ptr(x), ldx if ldx is not None else C_, _ptr(att), ldatt if ldatt is not None else C_, _ptr(stream),
                     _ptr(vec), _ptr(out2), ld2, _ptr(stats_out), M, C_, float(eps), float(eps_next), 1 if has_qkv else 0,
                     _ptr(q), _ptr(k), _ptr(vt), float(qscale), heads, hd, Tp, _nat.dtype_code(x.dtype))
        self._keep = (x, att, stream, vec, out2, stats_out, q, k, vt)
        self._fn = _nat.lib().lwdetr_vit_block

    def __call__(self, stream=None):
        rc = self._fn(*self.args, stream if stream is not None else _nat.stream_ptr())
        if rc:
            _nat.check(rc, "vit_block")


# ------------------------------------------------------------------------------------------------- row chains (chain.hip)
CHAIN_MIN_ROWS = 2048           # below this the separate launches spread over more CUs than M / 128 workgroups would


def enc_chain_supported(d, dtype, k5=0, ncls=91, nl=3, rows=None) -> bool:
    """Shapes / dtypes lwdetr_enc_chain is instantiated for; with ``rows`` the launch-plan choice (LWDETR_CHAIN=0/1 forces)."""
    ok = dtype in (torch.float16, torch.bfloat16) and ((d == 256 and k5 in (0, 640)) or (d == 384 and k5 == 0)) and ncls <= 96 and 1 <= nl <= 6
    if not ok or rows is None:
        return ok
    force = os.environ.get("LWDETR_CHAIN")
    if force in ("0", "1"):
        return force == "1"
    return rows >= CHAIN_MIN_ROWS


def chain_pieces(w, kslots=None):
    """(N, K) f32 weight, N % 32 == 0, K % 64 == 0 -> flat stream of N/32 * K/64 pieces of 4 KB (in 16-bit), tile-major: piece =
    32 output channels x 64 k-slots = 4 MFMA fragments of 1 KB in lane order (lane l = 32 h + i holds row i, slots 8 h .. 8 h + 7 of
    the fragment). ``kslots``: channel held by k-slot p (None = natural order: the operand rows come straight from memory;
    vb_kslot_channels(K): they are an accumulator tile handed on as a B operand)."""
    n, k = w.shape
    assert n % 32 == 0 and k % 64 == 0
    wk = w if kslots is None else w[:, kslots]
    return wk.reshape(n // 32, 32, k // 64, 4, 2, 8).permute(0, 2, 3, 4, 1, 5).reshape(-1)


def chain_pieces_split(w, kslots=None):
    """As chain_pieces for the channel-split form at K = 384 (lwdetr_row_chain, D = 384): inside every group of up to 4 tiles the pieces
    are ordered k-half-major - first the pieces [0, 3) of every tile of the group, then the pieces [3, 6) - because a step of that
    kernel is HALF a tile per wave (two steps of 4 x 3 pieces fit its 24-slot ring; 4 whole tiles = 24 pieces would not)."""
    n, k = w.shape
    assert n % 32 == 0 and k == 384
    t = chain_pieces(w, kslots).reshape(n // 32, 6, 2048)                 # tile, piece, elements
    out = []
    for g0 in range(0, n // 32, 4):
        grp = t[g0:g0 + 4]
        out += [grp[:, :3].reshape(-1), grp[:, 3:].reshape(-1)]
    return torch.cat(out)


def pack_enc_chain(d, dtype, w_enc, b_enc, g_enc, be_enc, w_cls, b_cls, w_val, b_val, cv2=None):
    """Host-side packing for lwdetr_enc_chain (f32 master tensors in) -> (stream of ``dtype``, vec f32).
    cv2 = (w2 (d, k5) BatchNorm-folded, b2 (d), ln_w (d), ln_b (d)) or None. Consumption order: cv2 | values | enc_output | class | 2 zero
    pieces. Operands that come from memory (cv2's input; without cv2 the `memory` rows) are in natural k order, operands handed on
    from an accumulator (after a LayerNorm) in k-slot order."""
    f = lambda t: t.detach().float().cpu()
    w_enc, b_enc, g_enc, be_enc, w_cls, b_cls, w_val, b_val = map(f, (w_enc, b_enc, g_enc, be_enc, w_cls, b_cls, w_val, b_val))
    perm = vb_kslot_channels(d)
    mem_slots = perm if cv2 is not None else None
    ncls = w_cls.shape[0]
    assert w_enc.shape == (d, d) and w_cls.shape[1] == d and ncls <= 96 and w_val.shape[1] == d and w_val.shape[0] % d == 0
    nl = w_val.shape[0] // d
    parts, vec = [], []
    if cv2 is not None:
        w2, b2, lw, lb = map(f, cv2)
        assert w2.shape[0] == d and w2.shape[1] % 64 == 0
        parts.append(chain_pieces(w2))
        vec += [b2, lw, lb]
    parts.append(chain_pieces(w_val, mem_slots))
    parts.append(chain_pieces(w_enc, mem_slots))
    wc = torch.zeros(96, d); wc[:ncls] = w_cls
    bc = torch.zeros(96); bc[:ncls] = b_cls
    parts.append(chain_pieces(wc, perm))
    parts.append(torch.zeros(2 * 2048))
    bv = torch.zeros(6 * d); bv[:nl * d] = b_val
    vec += [b_enc, g_enc, be_enc, bc, bv]
    vec = torch.cat(vec)
    nvec = (vec.numel() * 4 + 4095) // 4096 * 4096 // 4
    vec = torch.cat([vec, torch.zeros(nvec - vec.numel())]).contiguous()
    return torch.cat(parts).to(dtype).contiguous(), vec


class EncChainOp:
    """[C2f.cv2 + LayerNorm ->] memory -> value projections, enc_output + LayerNorm, class logits + row max: one launch (lwdetr_enc_chain)."""

    def __init__(self, inp, ld_in, k5, memory, om, cls, ld_cls, cls_max, values, rowvalid, notpad, stream, vec, *, M, d, npix, S, lsi,
                 total_rows, ncls, eps_p, eps_e):
        nl = len(values)
        assert stream.dtype == inp.dtype == om.dtype and vec.dtype == torch.float32 and cls_max.dtype == torch.float32
        assert stream.numel() * 2 == _nat.lib().lwdetr_enc_chain_pieces(d, k5, nl) * 4096, "stream size"
        assert vec.numel() == _nat.lib().lwdetr_enc_chain_vec_floats(d, k5), "vec size"
        self._vals = (C.c_void_p * nl)(*[v.data_ptr() for v in values])
        self.args = (_ptr(inp), ld_in, k5, _ptr(memory), _ptr(om), _ptr(cls), ld_cls, _ptr(cls_max), self._vals, nl, _ptr(rowvalid),
                     _ptr(notpad), _ptr(stream), _ptr(vec), M, d, npix, S, lsi, total_rows, ncls, float(eps_p), float(eps_e),
                     _nat.dtype_code(inp.dtype))
        self._keep = (inp, memory, om, cls, cls_max, values, rowvalid, notpad, stream, vec)
        self._fn = _nat.lib().lwdetr_enc_chain

    def __call__(self, stream=None):
        rc = self._fn(*self.args, stream if stream is not None else _nat.stream_ptr())
        if rc:
            _nat.check(rc, "enc_chain")


def row_chain_supported(d, dtype, k_in=None, res=False, qpos=False) -> bool:
    """Shapes lwdetr_row_chain is instantiated for (LWDETR_CHAIN=0 switches every chain off)."""
    if os.environ.get("LWDETR_CHAIN") == "0" or dtype not in (torch.float16, torch.bfloat16):
        return False
    k_in = k_in or d
    # The launch plan uses the chains at d = 256, k_in = d (channel-split form, 32 rows per workgroup: bs=1 0.925 -> 0.872 ms, B = 32
    # +3.8 % on LW-DETR-small). Implemented and tested but NOT in the plan (LWDETR_CHAIN_ALL=1 enables them): d = 384 - a step of the
    # split form is half a tile per wave there (12 MFMAs between two ring waits): out_proj + LN + offsets 49.6 vs 43.1 us as separate
    # launches, LW-DETR-large B = 32 4.15 k vs 4.39 k img/s, bs=1 unchanged; k_in = 2 d (ref_point_head) only exists in the row-per-wave
    # form, which runs 128 dependent MFMAs per stage on one wave (profiles/r4b_row_chain_forms.txt).
    all_forms = os.environ.get("LWDETR_CHAIN_ALL") == "1"
    if k_in == d and (d == 256 or (d == 384 and all_forms)):
        return res or not qpos
    return all_forms and d == 256 and k_in == 2 * d and not res and not qpos


class RowChainOp:
    """A chain of Linear stages over rows in one launch (lwdetr_row_chain). ``stages``: list of dicts
      dict(kind="full", w (D, K), b (D), relu=False, res=False, ln=(gamma, beta, eps) | None, out=tensor | None, ldo=None, addq=False)
      dict(kind="side", w (n, D), b (n), out=tensor, ldo)
    f32 master weights; the packed stream / vectors are built here (cached by the caller through ``packed=``)."""

    @staticmethod
    def pack(d, dtype, k_in, stages):
        perm = vb_kslot_channels(d)
        natural = True                  # operand rows straight from memory until the first FULL stage hands its accumulators on
        parts, vec = [], []
        f = lambda t: t.detach().float().cpu()
        pieces = chain_pieces_split if d == 384 else chain_pieces      # d = 384: only the channel-split form exists (k-half-major stream)
        for i, st in enumerate(stages):
            w, b = f(st["w"]), f(st["b"])
            if st["kind"] == "full":
                k = k_in if i == 0 else d
                assert w.shape == (d, k) and b.shape == (d,)
                parts.append(pieces(w, None if natural else perm))
                vec.append(b)
                if st.get("ln") is not None:
                    vec += [f(st["ln"][0]), f(st["ln"][1])]
                natural = False
            else:
                n = w.shape[0]
                nt = (n + 31) // 32
                assert w.shape[1] == d and b.shape == (n,)
                wp = torch.zeros(32 * nt, d); wp[:n] = w
                bp = torch.zeros(32 * nt); bp[:n] = b
                parts.append(pieces(wp, None if natural else perm))
                vec.append(bp)
        parts.append(torch.zeros(2 * 2048))
        vec = torch.cat(vec)
        nvec = (vec.numel() * 4 + 4095) // 4096 * 4096 // 4
        return torch.cat(parts).to(dtype).contiguous(), torch.cat([vec, torch.zeros(nvec - vec.numel())]).contiguous()

    def __init__(self, inp, ld_in, k_in, stages, stream, vec, *, M, d, res=None, ld_res=0, qpos=None, ld_q=0):
        assert stream.dtype == inp.dtype and vec.dtype == torch.float32 and len(stages) <= 6
        desc = _nat.ChainDesc()
        desc.inp, desc.ld_in, desc.k_in = _ptr(inp), ld_in, k_in
        desc.res, desc.ld_res, desc.qpos, desc.ld_q = _ptr(res), ld_res, _ptr(qpos), ld_q
        desc.wstream, desc.vec, desc.M, desc.D, desc.nst = _ptr(stream), _ptr(vec), M, d, len(stages)
        keep = [inp, res, qpos, stream, vec]
        for i, st in enumerate(stages):
            s = desc.st[i]
            if st["kind"] == "full":
                s.kind, s.n = _nat.CHAIN_FULL, d
                ln = st.get("ln")
                s.flags = ((_nat.CHAIN_RES if st.get("res") else 0) | (_nat.CHAIN_RELU if st.get("relu") else 0) | (_nat.CHAIN_LN if ln is not None else 0) |
                           (_nat.CHAIN_STORE if st.get("out") is not None else 0) | (_nat.CHAIN_ADDQ if st.get("addq") else 0))
                s.eps = float(ln[2]) if ln is not None else 0.0
                if st.get("out") is not None:
                    s.out, s.ldo = _ptr(st["out"]), st.get("ldo") or d
                    keep.append(st["out"])
            else:
                s.kind, s.n, s.flags, s.eps = _nat.CHAIN_SIDE, st["w"].shape[0] if "w" in st else st["n"], 0, 0.0
                s.out, s.ldo = _ptr(st["out"]), st["ldo"]
                keep.append(st["out"])
        lib = _nat.lib()
        assert stream.numel() * 2 == lib.lwdetr_row_chain_pieces(C.byref(desc)) * 4096, "row chain: stream size"
        assert vec.numel() == lib.lwdetr_row_chain_vec_floats(C.byref(desc)), "row chain: vec size"
        self.desc, self.dtype, self._keep = desc, _nat.dtype_code(inp.dtype), keep
        self._fn, self._ref = lib.lwdetr_row_chain, C.byref(desc)

    def __call__(self, stream=None):
        rc = self._fn(self._ref, self.dtype, stream if stream is not None else _nat.stream_ptr())
        if rc:
            _nat.check(rc, "row_chain")


class RawOp:
    """Generic pre-bound launch: ``fn(*args, stream)`` of the C ABI (keeps the tensors behind the pointers alive)."""

    def __init__(self, name, args, keep):
        self.name, self.args, self._keep = name, tuple(args), keep
        self._fn = getattr(_nat.lib(), name)

    def __call__(self, stream=None):
        rc = self._fn(*self.args, stream if stream is not None else _nat.stream_ptr())
        if rc:
            _nat.check(rc, self.name)

    def call_with(self, stream, repl):
        """Launch with some positional arguments replaced ({index: value}) - per-call output pointers."""
        args = list(self.args)
        for i, v in repl.items():
            args[i] = v
        rc = self._fn(*args, stream if stream is not None else _nat.stream_ptr())
        if rc:
            _nat.check(rc, self.name)


class MsdaFusedOp:
    def __init__(self, value, shapes, lsi, oa, ld_oa, logit_col, ref, vr, out, *, B, S, M, D, L, Q, P):
        assert ref.dtype == torch.float32 and vr.dtype == torch.float32
        assert shapes.dtype == torch.int64 and lsi.dtype == torch.int64
        self.args = (_ptr(value), _ptr(shapes), _ptr(lsi), _ptr(oa), ld_oa, logit_col, _ptr(ref), _ptr(vr),
                     _ptr(out), B, S, M, D, L, Q, P, _nat.dtype_code(value.dtype))
        self._keep = (value, shapes, lsi, oa, ref, vr, out)
        self._fn = _nat.lib().lwdetr_msda_fused_forward

    def __call__(self, stream=None):
        rc = self._fn(*self.args, stream if stream is not None else _nat.stream_ptr())
        if rc:
            _nat.check(rc, "msda_fused_forward")


# ------------------------------------------------------------------------------------------- one-shot wrappers
def linear(x, w, bias=None, act=ACT_NONE, res=None, gamma=None):
    """Plain fused linear on 2-D x (M,K) -> (M,N); convenience for tests and small call sites."""
    M, K = x.shape
    Nn = w.shape[0]
    out = torch.empty(M, Nn, dtype=x.dtype, device=x.device)
    b = None if bias is None else bias.float().contiguous()
    g = None if gamma is None else gamma.float().contiguous()
    s = seg(out, 0, Nn, ldo=Nn, bias=b, act=act, res=res, ldres=Nn if res is not None else 0, gamma=g)
    GemmOp(x, w.contiguous(), M, Nn, K, [s], keep=(b, g, res))()
    return out


def layernorm(x, gamma, beta, eps):
    M, C_ = x.shape
    out = torch.empty_like(x)
    LayerNormOp(x, gamma.float().contiguous(), beta.float().contiguous(), out, M, C_, eps)()
    return out


def attention_scale(hd: int) -> float:
    return hd ** -0.5 * LOG2E
