"""Launch-plan engine: turns an LW-DETR state dict into packed device weights and a flat list of pre-built kernel
launches (C ABI of ``liblwdetr_hip.so``) for one (batch, resolution, dtype).

Data layout (all activations are token-major / NHWC, dtype T = model dtype):
  * ViT stream  x[B*Tp, C]: tokens in the reference's window-major order (``models/backbone/vit.py:353-358``), every
    window padded to Twp = ceil4(h*w) rows so that all 8/16-byte vector accesses stay aligned (Tp = 16*Twp; no padding
    at 640x640 where h*w = 100). Window attention = 16 contiguous sequences per image, global attention = one.
  * Q, K (B, heads, Tp, hd), V^T (B, heads, hd, Tp): written directly by the QKV GEMM epilogue.
  * projector / memory: raster NHWC; ``memory`` is (B, S, d) with the levels concatenated - the projector's final
    LayerNorm writes straight into its level slice, so there is no flatten/transpose/cat (``transformer.py:199-222``).
Dead compute of the reference that is not executed (results unused): sine position embedding of
memory, head-averaged self-attention weights, per-call bicubic pos-embed resize (cached per resolution).
"""
import math
import os

import torch
import torch.nn.functional as F

from . import kernels as K
from .kernels import (A_CONV3x3, A_PATCH16, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SILU, OUT_DECONV2x2, OUT_HEADS,
                      OUT_HEADS_T, OUT_LINEAR, OUT_TOKMAP, AttnOp, GemmOp, LayerNormOp, MsdaFusedOp, seg, tok_layout)
from .configs import LEVEL_SCALE, VIT_SIZES


def _ceil4(n):
    return (n + 3) // 4 * 4


class PackedWeights:
    """Device-resident, kernel-ready copies of the parameters (weights in T, biases / scales in f32)."""

    def __init__(self, sd, cfg, device, dtype):
        self.device, self.dtype, self.cfg = device, dtype, cfg
        self.sd = sd
        self._cache = {}

    def w(self, key, fn=None, tag=""):
        """Weight matrix in T (optionally transformed by ``fn`` on the f32 master copy)."""
        ck = ("w", key, tag)
        if ck not in self._cache:
            t = self.sd[key].detach().float()
            if fn is not None:
                t = fn(t)
            self._cache[ck] = t.to(device=self.device, dtype=self.dtype).contiguous()
        return self._cache[ck]

    def f(self, key, fn=None, tag=""):
        """f32 vector (bias / LayerScale / norm parameters)."""
        ck = ("f", key, tag)
        if ck not in self._cache:
            t = self.sd[key].detach().float()
            if fn is not None:
                t = fn(t)
            self._cache[ck] = t.to(device=self.device, dtype=torch.float32).contiguous()
        return self._cache[ck]

    def custom(self, name, builder, dtype=None):
        ck = ("c", name)
        if ck not in self._cache:
            self._cache[ck] = builder().to(device=self.device, dtype=dtype or self.dtype).contiguous()
        return self._cache[ck]

    def custom_multi(self, name, builder):
        """Like ``custom`` for builders returning a tuple of tensors that already carry their final dtype."""
        ck = ("m", name)
        if ck not in self._cache:
            self._cache[ck] = tuple(t.to(device=self.device).contiguous() for t in builder())
        return self._cache[ck]

    # ---- conv + eval-BatchNorm folding (reference ConvX: conv(bias=False) -> BN -> act, projector.py:85-98)
    def convx_f32(self, prefix):
        """BatchNorm-folded conv weight (Cout, ky*kx*Cin) and bias, f32 masters on the parameters' device."""
        w = self.sd[prefix + ".conv.weight"].detach().float()
        g = self.sd[prefix + ".bn.weight"].detach().float()
        b = self.sd[prefix + ".bn.bias"].detach().float()
        mu = self.sd[prefix + ".bn.running_mean"].detach().float()
        var = self.sd[prefix + ".bn.running_var"].detach().float()
        s = g / torch.sqrt(var + 1e-5)
        w = w * s[:, None, None, None]
        return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1), b - mu * s      # k = tap*Cin + ci

    def convx(self, prefix):
        ck = ("convx", prefix)
        if ck not in self._cache:
            wk, bk = self.convx_f32(prefix)
            self._cache[ck] = (wk.to(device=self.device, dtype=self.dtype).contiguous(),
                               bk.to(device=self.device, dtype=torch.float32).contiguous())
        return self._cache[ck]


def abs_pos_winmajor(pos_embed, hp, wp, twp):
    """pos_embed (1, 1+14*14, C) -> (16*twp, C) f32: cls token dropped, bicubic resize (align_corners=False),
    window-major order, zero pad rows. Reference: models/backbone/vit.py:26-54, :353-358."""
    p = pos_embed[:, 1:].float()
    s = int(math.isqrt(p.shape[1]))
    c = p.shape[-1]
    if (s, s) != (hp, wp):
        p = F.interpolate(p.reshape(1, s, s, c).permute(0, 3, 1, 2), size=(hp, wp), mode="bicubic",
                          align_corners=False).permute(0, 2, 3, 1)
    else:
        p = p.reshape(1, hp, wp, c)
    h, w = hp // 4, wp // 4
    p = p.reshape(4, h, 4, w, c).permute(0, 2, 1, 3, 4).reshape(16, h * w, c)
    out = torch.zeros(16, twp, c, dtype=torch.float32, device=p.device)
    out[:, :h * w] = p
    return out.reshape(16 * twp, c)


def sine_embed(pos, dim):
    """(B, nq, 4) -> (B, nq, 4*dim) in order (y, x, w, h). Reference: models/transformer.py:42-68."""
    dim_t = torch.arange(dim, dtype=torch.float32, device=pos.device)
    dim_t = 10000 ** (2 * (dim_t // 2) / dim)
    out = []
    for idx in (1, 0, 2, 3):
        e = pos[:, :, idx, None] * (2 * math.pi) / dim_t
        out.append(torch.stack((e[:, :, 0::2].sin(), e[:, :, 1::2].cos()), dim=3).flatten(2))
    return torch.cat(out, dim=2)


def reparam(delta, ref):
    """Box re-parameterisation, no sigmoid (reference models/transformer.py:236-240, models/lwdetr.py:150-155)."""
    return torch.cat([delta[..., :2] * ref[..., 2:] + ref[..., :2], delta[..., 2:].exp() * ref[..., 2:]], -1)


class ForwardPlan:
    """All launches of one inference forward for fixed (B, H, W, dtype); buffers are allocated once."""

    def __init__(self, pw: PackedWeights, batch, height, width):
        cfg = pw.cfg
        self.pw, self.cfg = pw, cfg
        self.B, self.H, self.W = batch, height, width
        dev, T = pw.device, pw.dtype
        self.dev, self.T = dev, T
        if height % 64 or width % 64:
            raise AssertionError("image height and width must be multiples of 64 (vit.py:354)")
        C, heads = VIT_SIZES[cfg.encoder]
        self.C, self.heads, self.hd = C, heads, C // heads
        self.Hp, self.Wp = height // 16, width // 16
        h, w = self.Hp // 4, self.Wp // 4
        self.Tw, self.Twp = h * w, _ceil4(h * w)
        self.Tp = 16 * self.Twp
        self.rows = batch * self.Tp
        self.depth = cfg.vit_encoder_num_layers
        self.taps = sorted(i if i >= 0 else i + self.depth for i in cfg.out_feature_indexes)
        self.d = cfg.hidden_dim
        self.nq = cfg.num_queries
        self.L = len(cfg.projector_scale)
        self.level_hw = [(int(self.Hp * LEVEL_SCALE[s]), int(self.Wp * LEVEL_SCALE[s])) for s in cfg.projector_scale]
        self.lsi = [0]
        for (a, b) in self.level_hw[:-1]:
            self.lsi.append(self.lsi[-1] + a * b)
        self.S = sum(a * b for a, b in self.level_hw)
        if self.S < self.nq:
            raise RuntimeError(f"image too small: {self.S} memory tokens < {self.nq} queries")
        self.win_tok = tok_layout(True, self.Hp, self.Wp, self.Twp)
        self.ops_backbone, self.ops_enc, self.ops_sel, self.ops_dec = [], [], [], []
        self.debug = {}
        self.buffers = []            # every tensor the plan's kernels write (tests/test_gpu_chains.py restores / compares them)
        self._own = lambda t: (self.buffers.append(t), t)[1]
        # LWDETR_ARENA=1 carves the work buffers out of a few large allocations (256 MB chunks) instead of ~200 separate ones -
        # an experiment for the boxes on which the model process alone runs slow (profiles/r3f_box_spread.txt: suspected page
        # placement); on the usual boxes it measures the same or 0.5 % slower (12.50 vs 12.57 k img/s), so it is off by default.
        self._arena, self._arena_off = None, 0
        use_arena = os.environ.get("LWDETR_ARENA", "0") == "1"

        def zeros(*s, dt=None):
            dt = dt or T
            if len(s) == 1 and isinstance(s[0], (tuple, list)):
                s = tuple(s[0])
            n = 1
            for v in s:
                n *= int(v)
            nbytes = n * torch.empty(0, dtype=dt).element_size()
            if not use_arena or nbytes == 0:
                return self._own(torch.zeros(*s, dtype=dt, device=dev))
            chunk = 256 << 20
            need = (nbytes + 255) // 256 * 256
            if self._arena is None or self._arena_off + need > self._arena.numel():
                self._arena, self._arena_off = torch.empty(max(chunk, need), dtype=torch.uint8, device=dev), 0
            t = self._arena[self._arena_off:self._arena_off + nbytes].view(dt).view(*s)
            self._arena_off += need
            return self._own(t.zero_())

        self._z = zeros
        self._build_vit()
        self._build_projector()
        self._build_transformer()

    # ------------------------------------------------------------------------------------------------ ViT
    def _build_vit(self):
        pw, C, B, Tp, rows, heads, hd = self.pw, self.C, self.B, self.Tp, self.rows, self.heads, self.hd
        z, ops = self._z, self.ops_backbone
        pre = "backbone.0.encoder"
        self.images = None           # allocated only if the caller's tensor cannot be read in place
        self.x = z(rows, C)
        xn, att = z(rows, C), z(rows, C)
        hid = None if K.mlp_fused_supported(C, self.T, rows) else z(rows, 4 * C)
        q, k = z(B, heads, Tp, hd), z(B, heads, Tp, hd)
        self._vt_store = z(B * heads * hd * Tp + 8)          # 16 bytes of slack behind V^T (AttnDesc.vt_slack)
        vt = self._vt_store[:B * heads * hd * Tp].view(B, heads, hd, Tp)
        ntap = len(self.taps)
        self.taps_cat = z(rows, ntap * C)
        pos = pw.custom(f"pos.{self.Hp}x{self.Wp}", lambda: abs_pos_winmajor(pw.sd[pre + ".pos_embed"].detach().cpu(),
                                                                           self.Hp, self.Wp, self.Twp))
        qscale = K.attention_scale(hd)
        fused = K.mlp_fused_supported(C, self.T, rows)
        # LayerNorm folded into the following GEMM on the unfused path (the C = 768 model): built, validated - and worth NOTHING against the round-4
        # tree on the same box (xlarge 960 x 960 B = 16: 774.0 / 772.2 / 770.5 img/s round-4 tree, 774.0 / 771.2 / 774.2 this tree without the fold,
        # 774.9 / 772.9 / 773.0 with it): the folded epilogue's own kernel runs QKV 290 -> 310 us and fc1 372 -> 399 us, which is what the two
        # 37 -> 19 us statistics passes save (profiles/r5d_*, r5g_*). Opt-in: LWDETR_LN_FOLD=1 (rows >= 16 384: the large-tile kernel's shapes).
        # The folded epilogue exists in the 256 x 256 large-tile kernel only, which the library takes by itself from 16 384 rows: "1" folds where that
        # kernel will serve the launch, "2" whatever the row count (tests that force the large-tile kernel with lwdetr_gemm_tuning(2)); and only a
        # library built with -DLWDETR_EXPERIMENTS carries it (round 6) - otherwise the plan keeps LayerNorm + plain GEMM instead of failing at launch.
        lf_env = os.environ.get("LWDETR_LN_FOLD")
        ln_fold = ((not fused) and self.T != torch.float32 and C % 256 == 0 and (lf_env == "2" or (lf_env == "1" and rows >= 16384))
                   and bool(K._nat.lib().lwdetr_has_experiments()))
        ln_stats = torch.empty(2, rows, dtype=torch.float32, device=self.dev) if ln_fold else None
        self.ln_fold = ln_fold
        # (Statistics out of the epilogue of the GEMM that PRODUCES the rows were built in four forms and measured a tie at best against this
        # pass - and their code in the shared epilogue was part of what slowed every large-tile GEMM down: removed, profiles/r5d_*, r5g_*.)
        stats_op = lambda: K.RowStatsOp(self.x, ln_stats, rows, C, 1e-6)
        blk0_fused = fused and K.vit_block_supported(C, self.T, hd, rows) and rows % Tp == 0 and Tp % 8 == 0
        # round 4: patch embedding + position embedding + block 0's norm1 / QKV as ONE launch at the batch sizes of the block kernel
        self.stem_op = self.patch_op = None
        if (blk0_fused and os.environ.get("LWDETR_VIT_STEM", "1") != "0" and self.H == 16 * self.Hp and self.W == 16 * self.Wp
                and self.Hp % 4 == 0 and self.Wp % 4 == 0 and Tp == 16 * self.Twp and pos.dtype == self.T):
            b0 = pre + ".blocks.0"
            ss, sv = pw.custom_multi(pre + ".vitstem.packed", lambda: K.pack_vit_stem(
                pw.sd[pre + ".patch_embed.proj.weight"], pw.sd[pre + ".patch_embed.proj.bias"], pw.sd[b0 + ".attn.qkv.weight"],
                pw.sd[b0 + ".attn.q_bias"], pw.sd[b0 + ".attn.v_bias"], pw.sd[b0 + ".norm1.weight"], pw.sd[b0 + ".norm1.bias"], self.T))
            self.stem_op = K.VitStemOp(None, pos, self.x, ss, sv, B, self.Hp, self.Wp, self.Twp, C, 1e-6, q=q, k=k, vt=vt, qscale=qscale,
                                       heads=heads, hd=hd)
            ops.append(self.stem_op)
        else:
            wpe = pw.w(pre + ".patch_embed.proj.weight", lambda t: t.reshape(t.shape[0], -1))
            dummy_img = z(1, 8)
            ops.append(GemmOp(dummy_img, wpe, rows, C, 768, [
                seg(self.x, 0, C, ldo=C, bias=pw.f(pre + ".patch_embed.proj.bias"), res=pos, ldres=C, res_mod=Tp)],
                a_mode=A_PATCH16, a_tok=self.win_tok, img_h=self.H, img_w=self.W, keep=(pos,)))
            self.patch_op = ops[-1]
        for i in range(self.depth):
            blk = f"{pre}.blocks.{i}"
            window = i in self.cfg.window_block_indexes
            if i == 0 and self.stem_op is not None:
                pass                                     # q / k / v^T of block 0 come out of the stem launch
            elif i == 0 and blk0_fused and os.environ.get("LWDETR_VIT_QKV", "1") != "0":
                # block 0: norm1 + QKV in one launch (the QKV phase of the block kernel on its own; round 4)
                sq, vq = pw.custom_multi(blk + ".vitqkv.packed", lambda blk=blk: K.pack_vit_qkv(
                    pw.sd[blk + ".attn.qkv.weight"], pw.sd[blk + ".attn.q_bias"], pw.sd[blk + ".attn.v_bias"],
                    pw.sd[blk + ".norm1.weight"], pw.sd[blk + ".norm1.bias"], self.T))
                ops.append(K.VitQkvOp(self.x, sq, vq, rows, C, 1e-6, q=q, k=k, vt=vt, qscale=qscale, heads=heads, hd=hd, Tp=Tp))
            elif (i == 0 or not fused) and ln_fold:
                # round 5 (the unfused C = 768 path): norm1 folded into the QKV GEMM - row statistics only (half the LayerNorm's traffic),
                # the GEMM reads the raw rows, its epilogue applies (acc - mean colsum) rstd + b' (kernels.fold_layernorm)
                wq_, cs_, bq_ = pw.custom_multi(blk + ".qkv.lnfold", lambda blk=blk: K.fold_layernorm(
                    pw.sd[blk + ".attn.qkv.weight"], torch.cat([pw.sd[blk + ".attn.q_bias"].detach().float(), torch.zeros(C, device=self.dev),
                                                                pw.sd[blk + ".attn.v_bias"].detach().float()]),
                    pw.sd[blk + ".norm1.weight"], pw.sd[blk + ".norm1.bias"], self.T))
                ops.append(stats_op())
                ops.append(GemmOp(self.x, wq_, rows, 3 * C, C, [
                    seg(q, 0, C, mode=OUT_HEADS, bias=bq_[:C], scale=qscale, p0=Tp, p1=hd, p2=heads, ln_stats=ln_stats, ln_colsum=cs_[:C]),
                    seg(k, C, 2 * C, mode=OUT_HEADS, bias=bq_[C:2 * C], p0=Tp, p1=hd, p2=heads, ln_stats=ln_stats, ln_colsum=cs_[C:2 * C]),
                    seg(vt, 2 * C, 3 * C, mode=OUT_HEADS_T, bias=bq_[2 * C:], p0=Tp, p1=hd, p2=heads, ln_stats=ln_stats, ln_colsum=cs_[2 * C:])],
                    keep=(wq_, cs_, bq_)))
            elif i == 0 or not fused:
                # norm1 + QKV as separate launches (blocks > 0 get them chained into the previous block's MLP kernel)
                ops.append(LayerNormOp(self.x, pw.f(blk + ".norm1.weight"), pw.f(blk + ".norm1.bias"), xn, rows, C, 1e-6))
                ops.append(GemmOp(xn, pw.w(blk + ".attn.qkv.weight"), rows, 3 * C, C, [
                    seg(q, 0, C, mode=OUT_HEADS, bias=pw.f(blk + ".attn.q_bias"), scale=qscale, p0=Tp, p1=hd, p2=heads),
                    seg(k, C, 2 * C, mode=OUT_HEADS, p0=Tp, p1=hd, p2=heads),
                    seg(vt, 2 * C, 3 * C, mode=OUT_HEADS_T, bias=pw.f(blk + ".attn.v_bias"), p0=Tp, p1=hd, p2=heads)]))
            if window:
                ops.append(AttnOp(q, k, vt, att, B=B, heads=heads, hd=hd, Tp=Tp, ldo=C, seqs_per_img=16,
                                  seq_tok_stride=self.Twp, keys_per_seq=self.Twp, sub_stride=self.Twp,
                                  sub_len=self.Tw, kind=0, vt_slack=True))
            else:
                ops.append(AttnOp(q, k, vt, att, B=B, heads=heads, hd=hd, Tp=Tp, ldo=C, seqs_per_img=1,
                                  seq_tok_stride=Tp, keys_per_seq=Tp, sub_stride=self.Twp, sub_len=self.Tw, kind=1))
            if not fused:
                ops.append(GemmOp(att, pw.w(blk + ".attn.proj.weight"), rows, C, C, [
                    seg(self.x, 0, C, ldo=C, bias=pw.f(blk + ".attn.proj.bias"), gamma=pw.f(blk + ".gamma_1"),
                        res=self.x, ldres=C)]))
            tap_out = None
            if i in self.taps:
                j = self.taps.index(i)
                tap_out = self.taps_cat[:, j * C:]
            if fused and K.vit_block_supported(C, self.T, hd, rows) and self._vit_block_ok(blk):
                # round-3 kernel (BASELINE batch sizes): projection + MLP + the next block's norm1 / QKV, one launch
                sd_ = pw.sd
                qkv_src, nxt = None, {}
                if i + 1 < self.depth:
                    nb = f"{pre}.blocks.{i + 1}"
                    qkv_src = tuple(sd_[nb + k_] for k_ in (".attn.qkv.weight", ".attn.q_bias", ".attn.v_bias", ".norm1.weight", ".norm1.bias"))
                    nxt = dict(q=q, k=k, vt=vt, qscale=qscale, heads=heads, hd=hd, Tp=Tp)
                stream_w, vec = pw.custom_multi(blk + ".vitblock.packed", lambda blk=blk, qkv_src=qkv_src: K.pack_vit_block(
                    sd_[blk + ".attn.proj.weight"], sd_[blk + ".attn.proj.bias"], sd_[blk + ".gamma_1"],
                    sd_[blk + ".mlp.fc1.weight"], sd_[blk + ".mlp.fc1.bias"], sd_[blk + ".mlp.fc2.weight"], sd_[blk + ".mlp.fc2.bias"],
                    sd_[blk + ".gamma_2"], sd_[blk + ".norm2.weight"], sd_[blk + ".norm2.bias"], self.T, qkv=qkv_src))
                ops.append(K.VitBlockOp(self.x, att, stream_w, vec, rows, C, 1e-6, out2=tap_out, ld2=ntap * C, eps_next=1e-6, **nxt))
            elif fused:
                # one launch: x += gamma1 * proj(att); x += gamma2 * fc2(GELU(fc1(LN(x))))  (vit.py:206-218)
                w1f, b1f, w2c = pw.custom_multi(blk + ".mlp.packed", lambda blk=blk: K.pack_mlp_weights(
                    pw.sd[blk + ".mlp.fc1.weight"], pw.sd[blk + ".mlp.fc1.bias"], pw.sd[blk + ".mlp.fc2.weight"],
                    pw.sd[blk + ".norm2.weight"], pw.sd[blk + ".norm2.bias"], self.T, proj=True))
                nxt = {}
                if i + 1 < self.depth:      # chain norm1 + QKV of block i+1 onto the rows this kernel just produced
                    nb = f"{pre}.blocks.{i + 1}"
                    wq, bq = pw.custom_multi(nb + ".qkv.packed", lambda nb=nb: K.pack_qkv_weights(
                        pw.sd[nb + ".attn.qkv.weight"], pw.sd[nb + ".attn.q_bias"], pw.sd[nb + ".attn.v_bias"],
                        pw.sd[nb + ".norm1.weight"], pw.sd[nb + ".norm1.bias"], self.T))
                    nxt = dict(wqkv=wq, bqkv=bq, q=q, k=k, vt=vt, qscale=qscale, heads=heads, hd=hd, Tp=Tp)
                wp_ = pw.w(blk + ".attn.proj.weight")
                cls = K.MlpFusedOp
                if K.vit_block_few_supported(C, self.T, rows):
                    # round 6: the few-token kernel loads its weights straight from L2 as MFMA fragments - hand them over fragment-major (one
                    # contiguous KB per fragment instead of 16 half lines): 30 -> 22 us per block launch at one image
                    cls = K.VitBlockFewOp
                    w1f = pw.custom(blk + ".mlp.fc1.frag", lambda w1f=w1f: K.pack_frag16(w1f))
                    wp_ = pw.custom(blk + ".attn.proj.frag", lambda wp_=wp_: K.pack_frag16(wp_))
                    if "wqkv" in nxt:
                        nxt["wqkv"] = pw.custom(nb + ".qkv.frag", lambda wq=nxt["wqkv"]: K.pack_frag16(wq))
                ops.append(cls(self.x, w1f, b1f, w2c, pw.f(blk + ".mlp.fc2.bias"), pw.f(blk + ".gamma_2"), rows,
                               C, 1e-6, out2=tap_out, ld2=ntap * C, att=att, ldatt=C,
                               wp=wp_, bp=pw.f(blk + ".attn.proj.bias"),
                               gamma1=pw.f(blk + ".gamma_1"), eps_next=1e-6, **nxt))
            else:
                if ln_fold:        # norm2 folded into fc1 (see norm1 above)
                    w1_, cs1_, b1_ = pw.custom_multi(blk + ".fc1.lnfold", lambda blk=blk: K.fold_layernorm(
                        pw.sd[blk + ".mlp.fc1.weight"], pw.sd[blk + ".mlp.fc1.bias"], pw.sd[blk + ".norm2.weight"], pw.sd[blk + ".norm2.bias"], self.T))
                    ops.append(stats_op())
                    ops.append(GemmOp(self.x, w1_, rows, 4 * C, C, [
                        seg(hid, 0, 4 * C, ldo=4 * C, bias=b1_, act=ACT_GELU, ln_stats=ln_stats, ln_colsum=cs1_)], keep=(w1_, cs1_, b1_)))
                else:
                    ops.append(LayerNormOp(self.x, pw.f(blk + ".norm2.weight"), pw.f(blk + ".norm2.bias"), xn, rows, C, 1e-6))
                    ops.append(GemmOp(xn, pw.w(blk + ".mlp.fc1.weight"), rows, 4 * C, C, [
                        seg(hid, 0, 4 * C, ldo=4 * C, bias=pw.f(blk + ".mlp.fc1.bias"), act=ACT_GELU)]))
                ops.append(GemmOp(hid, pw.w(blk + ".mlp.fc2.weight"), rows, C, 4 * C, [
                    seg(self.x, 0, C, ldo=C, bias=pw.f(blk + ".mlp.fc2.bias"), gamma=pw.f(blk + ".gamma_2"), res=self.x,
                        ldres=C, out2=tap_out, ld2=ntap * C)], keep=(tap_out,)))

    def _vit_block_ok(self, blk):
        """lwdetr_vit_block divides by the LayerScale vectors: blocks with (near-)zero entries stay on lwdetr_mlp_fused."""
        sd_ = self.pw.sd
        return min(sd_[blk + ".gamma_1"].detach().abs().min().item(), sd_[blk + ".gamma_2"].detach().abs().min().item()) >= K._VB_MIN_GAMMA

    # ------------------------------------------------------------------------------------------ projector
    def _convx_1x1(self, prefix, A, lda, M, cin, out_seg_fn, a_ptr_off=0):
        w, b = self.pw.convx(prefix)
        act = ACT_SILU if ".stages." in prefix else ACT_RELU
        return GemmOp(A, w, M, w.shape[0], cin, [out_seg_fn(b, act, w.shape[0])], lda=lda, keep=(w, b))

    def _build_projector(self):
        pw, cfg, C, B, d = self.pw, self.cfg, self.C, self.B, self.d
        z, ops = self._z, self.ops_backbone
        ntap = len(self.taps)
        pre = "backbone.0.projector"
        self.memory = z(B * self.S, d)
        self.level_feats = []
        # round 4: everything between the C2f block and the two-stage top-k as ONE launch (lwdetr_enc_chain, built in
        # _build_transformer); with one level and d = 256 the chain also takes the projector's cv2 + LayerNorm in front
        self.use_chain = K.enc_chain_supported(d, self.T, ncls=pw.sd["class_embed.weight"].shape[0], nl=cfg.dec_layers, rows=B * self.S)
        self.chain_front = None
        for li, name in enumerate(cfg.projector_scale):
            scale = LEVEL_SCALE[name]
            hl, wl = self.level_hw[li]
            npix = hl * wl
            M = B * npix
            ras = tok_layout(False, hl, wl, 0)
            st = f"{pre}.stages.{li}.0"
            c = d // 2
            ycat = z(M, 5 * c)
            w1, b1 = pw.convx(st + ".cv1")
            if scale == 1.0:
                # C2f.cv1 consumes the window-major tap concat and un-windows in its epilogue
                ops.append(GemmOp(self.taps_cat, w1, self.rows, 2 * c, ntap * C, [
                    seg(ycat, 0, 2 * c, mode=OUT_TOKMAP, ldo=5 * c, bias=b1, act=ACT_SILU, in_tok=self.win_tok,
                        out_tok=ras, out_batch_stride=npix * 5 * c)]))
            else:
                if scale == 2.0:
                    c_out = C // 4 if C > 512 else C // 2
                    cat = z(M, ntap * c_out)
                    for j in range(ntap):
                        sp = f"{pre}.stages_sampling.{li}.{j}"
                        src, lda, kin = self.taps_cat[:, j * C:], ntap * C, C
                        di = 0
                        if C > 512:
                            wx, bx = pw.convx(sp + ".0")
                            tmp = z(self.rows, C // 2)
                            ops.append(GemmOp(src, wx, self.rows, C // 2, C, [
                                seg(tmp, 0, C // 2, ldo=C // 2, bias=bx, act=ACT_RELU)], lda=lda, keep=(src,)))
                            src, lda, kin, di = tmp, C // 2, C // 2, 1
                        wd = pw.w(f"{sp}.{di}.weight", lambda t: t.permute(2, 3, 1, 0).reshape(-1, t.shape[0]))
                        bd = pw.f(f"{sp}.{di}.bias", lambda t: t.repeat(4))
                        dst = cat[:, j * c_out:]
                        ops.append(GemmOp(src, wd, self.rows, 4 * c_out, kin, [
                            seg(dst, 0, 4 * c_out, mode=OUT_DECONV2x2, ldo=ntap * c_out, bias=bd, p0=c_out,
                                in_tok=self.win_tok, out_tok=ras, out_batch_stride=npix * ntap * c_out)],
                            lda=lda, keep=(src, dst)))
                    kcat = ntap * c_out
                else:   # 0.5: ConvX(C, C, 3, stride 2), ReLU
                    cat = z(M, ntap * C)
                    for j in range(ntap):
                        wx, bx = pw.convx(f"{pre}.stages_sampling.{li}.{j}.0")
                        dst = cat[:, j * C:]
                        ops.append(GemmOp(self.taps_cat, wx, M, C, 9 * C, [
                            seg(dst, 0, C, ldo=ntap * C, bias=bx, act=ACT_RELU)], lda=ntap * C, a_mode=A_CONV3x3,
                            a_tok=self.win_tok, conv_cin=C, conv_stride=2, a_col0=j * C, conv_hout=hl, conv_wout=wl,
                            keep=(dst,)))
                    kcat = ntap * C
                ops.append(GemmOp(cat, w1, M, 2 * c, kcat, [seg(ycat, 0, 2 * c, ldo=5 * c, bias=b1, act=ACT_SILU)]))
            tmp = z(M, c)
            # round 6: at one or two images the 3x3 convolutions run on the few-row kernel (fragment-major weights straight from L2, no LDS ring)
            few = K.gemm_few_supported(self.T, M, A_CONV3x3, c)
            conv_cls = K.GemmFewOp if few else GemmOp
            for m in range(3):
                wa, ba = pw.convx(f"{st}.m.{m}.cv1")
                wb, bb = pw.convx(f"{st}.m.{m}.cv2")
                if few:
                    wa = pw.custom(f"{st}.m.{m}.cv1.frag", lambda wa=wa: K.pack_frag16(wa))
                    wb = pw.custom(f"{st}.m.{m}.cv2.frag", lambda wb=wb: K.pack_frag16(wb))
                dst = ycat[:, (2 + m) * c:]
                ops.append(conv_cls(ycat, wa, M, c, 9 * c, [seg(tmp, 0, c, ldo=c, bias=ba, act=ACT_SILU)], lda=5 * c,
                                    a_mode=A_CONV3x3, a_tok=ras, conv_cin=c, conv_stride=1, a_col0=(1 + m) * c,
                                    conv_hout=hl, conv_wout=wl))
                ops.append(conv_cls(tmp, wb, M, c, 9 * c, [seg(dst, 0, c, ldo=5 * c, bias=bb, act=ACT_SILU)], lda=c,
                                    a_mode=A_CONV3x3, a_tok=ras, conv_cin=c, conv_stride=1, a_col0=0, conv_hout=hl,
                                    conv_wout=wl, keep=(dst,)))
            if self.use_chain and self.L == 1 and K.enc_chain_supported(d, self.T, k5=5 * c) and os.environ.get("LWDETR_CHAIN_FRONT", "1") != "0":
                self.chain_front = dict(ycat=ycat, k5=5 * c, cv2=st + ".cv2", ln=f"{pre}.stages.{li}.1", npix=npix, lsi=self.lsi[li], M=M)
                continue
            w2, b2 = pw.convx(st + ".cv2")
            zf = z(M, d)
            ops.append(GemmOp(ycat, w2, M, d, 5 * c, [seg(zf, 0, d, ldo=d, bias=b2, act=ACT_SILU)]))
            ops.append(LayerNormOp(zf, pw.f(f"{pre}.stages.{li}.1.weight"), pw.f(f"{pre}.stages.{li}.1.bias"),
                                   self.memory, M, d, 1e-6, rows_per_batch=npix, out_batch_rows=self.S,
                                   out_row_offset=self.lsi[li]))

    # ---------------------------------------------------------------------------------------- transformer
    def _build_transformer(self):
        pw, cfg, B, d, S, nq, L = self.pw, self.cfg, self.B, self.d, self.S, self.nq, self.L
        z = self._z
        t = "transformer"
        dev = self.dev
        self.shapes_t = torch.tensor(self.level_hw, dtype=torch.int64, device=dev)
        self.lsi_t = torch.tensor(self.lsi, dtype=torch.int64, device=dev)
        # ---- encoder-side (all S tokens): enc_output Linear+LN, class logits; value_proj of all decoder layers
        self.rowvalid = self._own(torch.ones(B * S, dtype=torch.uint8, device=dev))
        self.notpad = self._own(torch.ones(B * S, dtype=torch.uint8, device=dev))
        self.om = z(B * S, d)
        self.ncls = pw.sd["class_embed.weight"].shape[0]
        self.ldc = _ceil4(self.ncls)
        nl = cfg.dec_layers
        self.ldc_enc = 96 if self.use_chain else self.ldc
        self.enc_cls = z(B * S, self.ldc_enc)
        self.cls_max = self._own(torch.zeros(B, S, dtype=torch.float32, device=dev))
        ops = self.ops_enc
        if self.use_chain:
            self.values = [z(B * S, d) for _ in range(nl)]
            fr = self.chain_front
            sdv = pw.sd

            def build():
                cv2 = None
                if fr is not None:
                    w2, b2 = pw.convx_f32(fr["cv2"])
                    cv2 = (w2, b2, sdv[fr["ln"] + ".weight"], sdv[fr["ln"] + ".bias"])
                return K.pack_enc_chain(
                    d, self.T, sdv[f"{t}.enc_output.0.weight"], sdv[f"{t}.enc_output.0.bias"], sdv[f"{t}.enc_output_norm.0.weight"],
                    sdv[f"{t}.enc_output_norm.0.bias"], sdv[f"{t}.enc_out_class_embed.0.weight"], sdv[f"{t}.enc_out_class_embed.0.bias"],
                    torch.cat([sdv[f"{t}.decoder.layers.{i}.cross_attn.value_proj.weight"].detach().float() for i in range(nl)], 0),
                    torch.cat([sdv[f"{t}.decoder.layers.{i}.cross_attn.value_proj.bias"].detach().float() for i in range(nl)], 0), cv2=cv2)

            stream_w, vec = pw.custom_multi(f"{t}.enc_chain.packed.{'front' if fr else 'plain'}", build)
            if fr is not None:
                ops.append(K.EncChainOp(fr["ycat"], fr["k5"], fr["k5"], self.memory, self.om, self.enc_cls, self.ldc_enc, self.cls_max,
                                        self.values, self.rowvalid, self.notpad, stream_w, vec, M=fr["M"], d=d, npix=fr["npix"], S=S,
                                        lsi=fr["lsi"], total_rows=B * S, ncls=self.ncls, eps_p=1e-6, eps_e=1e-5))
            else:
                ops.append(K.EncChainOp(self.memory, d, 0, None, self.om, self.enc_cls, self.ldc_enc, self.cls_max, self.values,
                                        self.rowvalid, self.notpad, stream_w, vec, M=B * S, d=d, npix=S, S=S, lsi=0, total_rows=B * S,
                                        ncls=self.ncls, eps_p=1e-6, eps_e=1e-5))
        else:
            e1 = z(B * S, d)
            ops.append(GemmOp(self.memory, pw.w(f"{t}.enc_output.0.weight"), B * S, d, d, [
                seg(e1, 0, d, ldo=d, bias=pw.f(f"{t}.enc_output.0.bias"), rowmask=self.rowvalid)]))
            ops.append(LayerNormOp(e1, pw.f(f"{t}.enc_output_norm.0.weight"), pw.f(f"{t}.enc_output_norm.0.bias"),
                                   self.om, B * S, d, 1e-5))
            ops.append(GemmOp(self.om, pw.w(f"{t}.enc_out_class_embed.0.weight"), B * S, self.ncls, d, [
                seg(self.enc_cls, 0, self.ncls, ldo=self.ldc, bias=pw.f(f"{t}.enc_out_class_embed.0.bias"))]))
        if not self.use_chain:
            wv = pw.custom("value_proj_all.w", lambda: torch.cat(
                [pw.sd[f"{t}.decoder.layers.{i}.cross_attn.value_proj.weight"].detach().float() for i in range(nl)], 0))
            bv = pw.custom("value_proj_all.b", lambda: torch.cat(
                [pw.sd[f"{t}.decoder.layers.{i}.cross_attn.value_proj.bias"].detach().float() for i in range(nl)], 0),
                dtype=torch.float32)
            self.values = [z(B * S, d) for _ in range(nl)]
            vsegs = [seg(self.values[i], i * d, (i + 1) * d, ldo=d, bias=bv[i * d:], rowmask=self.notpad, rowmask_after=True)
                     for i in range(nl)]
            for g0 in range(0, nl, 3):
                grp = vsegs[g0:g0 + 3]
                for s_ in grp:
                    s_.n_begin -= g0 * d
                    s_.n_end -= g0 * d
                ops.append(GemmOp(self.memory, wv[g0 * d:], B * S, len(grp) * d, d, grp, keep=(wv, bv)))
        # ---- selected queries: bbox MLP of the two-stage head on the nq gathered rows only (row-wise op)
        self.om_sel = z(B * nq, d)
        s1, s2 = z(B * nq, d), z(B * nq, d)
        self.enc_delta = z(B * nq, 4)
        ops = self.ops_sel
        be = f"{t}.enc_out_bbox_embed.0.layers"
        sdw = pw.sd
        chain_plain = K.row_chain_supported(d, self.T)                                  # round 4: rows that never mix -> one launch
        chain_res = K.row_chain_supported(d, self.T, res=True)
        chain_q = K.row_chain_supported(d, self.T, res=True, qpos=True) and os.environ.get("LWDETR_DEC_QPOS", "1") != "0"
        chain_k2 = K.row_chain_supported(d, self.T, k_in=2 * d)

        def row_chain(name, inp, ld_in, k_in, M, stages, **kw):
            """stages: RowChainOp stage dicts with weights / biases given as state-dict keys (or f32 tensors)."""
            g = lambda v: sdw[v] if isinstance(v, str) else v
            for st in stages:
                st["w"], st["b"] = g(st["w"]), g(st["b"])
                if st.get("ln") is not None:
                    st["ln"] = (g(st["ln"][0]), g(st["ln"][1]), st["ln"][2])
            stream_w, vec = pw.custom_multi(f"row_chain.{name}", lambda: K.RowChainOp.pack(d, self.T, k_in, stages))
            return K.RowChainOp(inp, ld_in, k_in, stages, stream_w, vec, M=M, d=d, **kw)

        if chain_plain:
            ops.append(row_chain("enc_bbox", self.om_sel, d, d, B * nq, [
                dict(kind="full", w=be + ".0.weight", b=be + ".0.bias", relu=True),
                dict(kind="full", w=be + ".1.weight", b=be + ".1.bias", relu=True),
                dict(kind="side", w=be + ".2.weight", b=be + ".2.bias", out=self.enc_delta, ldo=4)]))
        else:
            ops.append(GemmOp(self.om_sel, pw.w(be + ".0.weight"), B * nq, d, d, [seg(s1, 0, d, ldo=d, bias=pw.f(be + ".0.bias"), act=ACT_RELU)]))
            ops.append(GemmOp(s1, pw.w(be + ".1.weight"), B * nq, d, d, [seg(s2, 0, d, ldo=d, bias=pw.f(be + ".1.bias"), act=ACT_RELU)]))
            ops.append(GemmOp(s2, pw.w(be + ".2.weight"), B * nq, 4, d, [seg(self.enc_delta, 0, 4, ldo=4, bias=pw.f(be + ".2.bias"))]))
        # ---- decoder
        ops = self.ops_dec
        M, D = cfg.ca_nheads, d // cfg.ca_nheads
        P = cfg.dec_n_points
        sa_h, sa_hd = cfg.sa_nheads, d // cfg.sa_nheads
        rq = B * nq
        self.xdec = z(rq, d)
        self.sine = z(rq, 2 * d)
        self.qpos = z(rq, d)
        self.ref = self._own(torch.zeros(B, nq, 4, dtype=torch.float32, device=dev))
        self.vr = self._own(torch.ones(B, L, 2, dtype=torch.float32, device=dev))
        self.hs = z(nl, rq, d)
        r1 = z(rq, d)
        rp = f"{t}.decoder.ref_point_head.layers"
        if chain_k2:
            ops.append(row_chain("ref_point_head", self.sine, 2 * d, 2 * d, rq, [
                dict(kind="full", w=rp + ".0.weight", b=rp + ".0.bias", relu=True),
                dict(kind="full", w=rp + ".1.weight", b=rp + ".1.bias", out=self.qpos, ldo=d)]))
        else:
            ops.append(GemmOp(self.sine, pw.w(rp + ".0.weight"), rq, d, 2 * d, [seg(r1, 0, d, ldo=d, bias=pw.f(rp + ".0.bias"), act=ACT_RELU)]))
            ops.append(GemmOp(r1, pw.w(rp + ".1.weight"), rq, d, d, [seg(self.qpos, 0, d, ldo=d, bias=pw.f(rp + ".1.bias"))]))
        qd, kd, vtd = z(B, sa_h, nq, sa_hd), z(B, sa_h, nq, sa_hd), z(B, sa_h, sa_hd, nq)
        attd, y, ca = z(rq, d), z(rq, d), z(rq, d)
        # decoder FFN: two launches with the hidden activation on chip (lwdetr_ffn_partial / _finish) unless
        # LWDETR_FFN_FUSED=0 (three launches: linear1 + ReLU, linear2 + residual, LayerNorm chain)
        ffn_fused = K.ffn_fused_supported(d, cfg.dim_feedforward, self.T) and os.environ.get("LWDETR_FFN_FUSED", "1") != "0"
        ffn = None if ffn_fused else z(rq, cfg.dim_feedforward)
        ffn_scratch = None
        lp3 = M * L * P * 3
        ld_oa = _ceil4(lp3)
        oa = z(rq, ld_oa)
        self.dec_bufs = dict(xdec=self.xdec, qpos=self.qpos, q=qd, k=kd, vt=vtd, att=attd, y=y, ca=ca, oa=oa, hs=self.hs)   # tools/determinism_probe.py
        # query_pos is the same in every layer (lite_refpoint_refine: the reference points are not refined between layers,
        # transformer.py:300-330), and it only ever enters a layer through (x + query_pos) W: the products query_pos Wq^T,
        # query_pos Wk^T and query_pos [W_offsets; W_weights]^T of ALL layers come out of ONE GEMM up front, and each layer adds
        # its slice in the epilogue of x W (the `res` term; the q slice is pre-multiplied by the attention scale through `gamma`).
        # Per layer that merges the q / k and v projections into one launch and takes the A + A2 operand sum (no LDS-DMA path)
        # out of the two GEMMs that had it. LWDETR_DEC_QPOS=0 keeps the round-2 plan (A2 = query_pos).
        qpos_pre = os.environ.get("LWDETR_DEC_QPOS", "1") != "0" and ld_oa % 8 == 0 and (2 * d) % 128 == 0
        if qpos_pre:
            s_att = K.attention_scale(sa_hd)
            g_pre = pw.custom(f"{t}.decoder.qpos_pre.gamma", lambda: torch.cat(
                [torch.full((d,), s_att), torch.ones(d)] * nl), dtype=torch.float32)
            if chain_q:      # the layer-front chain forms tgt + query_pos itself (CF_ADDQ): only the q / k products are precomputed
                w_pre = pw.custom(f"{t}.decoder.qpos_pre.w.qk", lambda: torch.cat(
                    [pw.sd[f"{t}.decoder.layers.{i}.self_attn.in_proj_weight"].detach().float()[:2 * d] for i in range(nl)], 0))
                pqk, poa = z(rq, nl * 2 * d), None
                ops.append(GemmOp(self.qpos, w_pre, rq, nl * 2 * d, d, [seg(pqk, 0, nl * 2 * d, ldo=nl * 2 * d, gamma=g_pre)]))
            else:
                w_pre = pw.custom(f"{t}.decoder.qpos_pre.w", lambda: torch.cat(
                    [pw.sd[f"{t}.decoder.layers.{i}.self_attn.in_proj_weight"].detach().float()[:2 * d] for i in range(nl)] +
                    [F.pad(torch.cat([pw.sd[f"{t}.decoder.layers.{i}.cross_attn.sampling_offsets.weight"].detach().float(),
                                      pw.sd[f"{t}.decoder.layers.{i}.cross_attn.attention_weights.weight"].detach().float()], 0),
                           (0, 0, 0, ld_oa - lp3)) for i in range(nl)], 0))
                pqk, poa = z(rq, nl * 2 * d), z(rq, nl * ld_oa)
                ops.append(GemmOp(self.qpos, w_pre, rq, nl * (2 * d + ld_oa), d, [
                    seg(pqk, 0, nl * 2 * d, ldo=nl * 2 * d, gamma=g_pre),
                    seg(poa, nl * 2 * d, nl * (2 * d + ld_oa), ldo=nl * ld_oa)]))
        for li in range(nl):
            lay = f"{t}.decoder.layers.{li}"
            ipw, ipb = lay + ".self_attn.in_proj_weight", lay + ".self_attn.in_proj_bias"
            if qpos_pre:
                ops.append(GemmOp(self.xdec, pw.w(ipw), rq, 3 * d, d, [
                    seg(qd, 0, d, mode=OUT_HEADS, bias=pw.f(ipb, lambda b_: b_[:d], "q"), scale=s_att, p0=nq, p1=sa_hd, p2=sa_h,
                        res=pqk[:, li * 2 * d:], ldres=nl * 2 * d),
                    seg(kd, d, 2 * d, mode=OUT_HEADS, bias=pw.f(ipb, lambda b_: b_[d:2 * d], "k"), p0=nq, p1=sa_hd, p2=sa_h,
                        res=pqk[:, li * 2 * d + d:], ldres=nl * 2 * d),
                    seg(vtd, 2 * d, 3 * d, mode=OUT_HEADS_T, bias=pw.f(ipb, lambda b_: b_[2 * d:], "v"), p0=nq, p1=sa_hd, p2=sa_h)],
                    keep=(pqk,)))
            else:
                ops.append(GemmOp(self.xdec, pw.w(ipw, lambda w_: w_[:2 * d], "qk"), rq, 2 * d, d, [
                    seg(qd, 0, d, mode=OUT_HEADS, bias=pw.f(ipb, lambda b_: b_[:d], "q"), scale=K.attention_scale(sa_hd),
                        p0=nq, p1=sa_hd, p2=sa_h),
                    seg(kd, d, 2 * d, mode=OUT_HEADS, bias=pw.f(ipb, lambda b_: b_[d:2 * d], "k"), p0=nq, p1=sa_hd, p2=sa_h)],
                    A2=self.qpos))
                ops.append(GemmOp(self.xdec, pw.w(ipw, lambda w_: w_[2 * d:], "v"), rq, d, d, [
                    seg(vtd, 0, d, mode=OUT_HEADS_T, bias=pw.f(ipb, lambda b_: b_[2 * d:], "v"), p0=nq, p1=sa_hd, p2=sa_h)]))
            ops.append(AttnOp(qd, kd, vtd, attd, B=B, heads=sa_h, hd=sa_hd, Tp=nq, ldo=d, seqs_per_img=1,
                              seq_tok_stride=nq, keys_per_seq=nq, sub_stride=nq, sub_len=nq, kind=2))
            ca_p = lay + ".cross_attn"
            front_chain = chain_q and qpos_pre
            if front_chain:
                # self_attn.out_proj + residual + norm1 -> tgt (stored), tgt + query_pos -> sampling_offsets | attention_weights: one launch
                ops.append(row_chain(f"dec{li}.front", attd, d, d, rq, [
                    dict(kind="full", w=lay + ".self_attn.out_proj.weight", b=lay + ".self_attn.out_proj.bias", res=True,
                         ln=(lay + ".norm1.weight", lay + ".norm1.bias", 1e-5), out=self.xdec, ldo=d, addq=True),
                    dict(kind="side", w=torch.cat([sdw[ca_p + ".sampling_offsets.weight"].detach().float(), sdw[ca_p + ".attention_weights.weight"].detach().float()], 0),
                         b=torch.cat([sdw[ca_p + ".sampling_offsets.bias"].detach().float(), sdw[ca_p + ".attention_weights.bias"].detach().float()], 0),
                         out=oa, ldo=ld_oa)], res=self.xdec, ld_res=d, qpos=self.qpos, ld_q=d))
            elif chain_res:
                ops.append(row_chain(f"dec{li}.front_noq", attd, d, d, rq, [
                    dict(kind="full", w=lay + ".self_attn.out_proj.weight", b=lay + ".self_attn.out_proj.bias", res=True,
                         ln=(lay + ".norm1.weight", lay + ".norm1.bias", 1e-5), out=self.xdec, ldo=d)], res=self.xdec, ld_res=d))
            else:
                ops.append(GemmOp(attd, pw.w(lay + ".self_attn.out_proj.weight"), rq, d, d, [
                    seg(y, 0, d, ldo=d, bias=pw.f(lay + ".self_attn.out_proj.bias"), res=self.xdec, ldres=d)]))
                ops.append(LayerNormOp(y, pw.f(lay + ".norm1.weight"), pw.f(lay + ".norm1.bias"), self.xdec, rq, d, 1e-5))
            w_oa = pw.custom(ca_p + ".oa.w", lambda ca_p=ca_p: torch.cat(
                [pw.sd[ca_p + ".sampling_offsets.weight"].detach().float(),
                 pw.sd[ca_p + ".attention_weights.weight"].detach().float()], 0))
            b_oa = pw.custom(ca_p + ".oa.b", lambda ca_p=ca_p: torch.cat(
                [pw.sd[ca_p + ".sampling_offsets.bias"].detach().float(),
                 pw.sd[ca_p + ".attention_weights.bias"].detach().float()], 0), dtype=torch.float32)
            if front_chain:
                pass
            elif qpos_pre:
                ops.append(GemmOp(self.xdec, w_oa, rq, lp3, d, [seg(oa, 0, lp3, ldo=ld_oa, bias=b_oa, res=poa[:, li * ld_oa:],
                                                                     ldres=nl * ld_oa)], keep=(poa,)))
            else:
                ops.append(GemmOp(self.xdec, w_oa, rq, lp3, d, [seg(oa, 0, lp3, ldo=ld_oa, bias=b_oa)], A2=self.qpos))
            ops.append(MsdaFusedOp(self.values[li], self.shapes_t, self.lsi_t, oa, ld_oa, M * L * P * 2, self.ref,
                                   self.vr, ca, B=B, S=S, M=M, D=D, L=L, Q=nq, P=P))
            if chain_res:
                ops.append(row_chain(f"dec{li}.back", ca, d, d, rq, [
                    dict(kind="full", w=ca_p + ".output_proj.weight", b=ca_p + ".output_proj.bias", res=True,
                         ln=(lay + ".norm2.weight", lay + ".norm2.bias", 1e-5), out=self.xdec, ldo=d)], res=self.xdec, ld_res=d))
            else:
                ops.append(GemmOp(ca, pw.w(ca_p + ".output_proj.weight"), rq, d, d, [
                    seg(y, 0, d, ldo=d, bias=pw.f(ca_p + ".output_proj.bias"), res=self.xdec, ldres=d)]))
                ops.append(LayerNormOp(y, pw.f(lay + ".norm2.weight"), pw.f(lay + ".norm2.bias"), self.xdec, rq, d, 1e-5))
            n3 = (pw.f(lay + ".norm3.weight"), pw.f(lay + ".norm3.bias"), 1e-5, self.xdec,
                  pw.f(f"{t}.decoder.norm.weight"), pw.f(f"{t}.decoder.norm.bias"), 1e-5, self.hs[li], rq, d)
            if ffn_fused:
                w1, b1, w2c = pw.custom_multi(lay + ".ffn.packed", lambda lay=lay: K.pack_mlp_weights(
                    pw.sd[lay + ".linear1.weight"], pw.sd[lay + ".linear1.bias"], pw.sd[lay + ".linear2.weight"], None, None,
                    self.T))
                if ffn_scratch is None:      # one scratch for all layers (they run back to back on the plan's stream)
                    ffn_scratch = torch.empty(K.ffn_partial_floats(rq, d, cfg.dim_feedforward, self.T), dtype=torch.float32, device=dev)
                ops.append(K.FfnOp(self.xdec, w1, b1, w2c, pw.f(lay + ".linear2.bias"), *n3, partial=ffn_scratch))
                continue
            ops.append(GemmOp(self.xdec, pw.w(lay + ".linear1.weight"), rq, cfg.dim_feedforward, d, [
                seg(ffn, 0, cfg.dim_feedforward, ldo=cfg.dim_feedforward, bias=pw.f(lay + ".linear1.bias"), act=ACT_RELU)]))
            ops.append(GemmOp(ffn, pw.w(lay + ".linear2.weight"), rq, d, cfg.dim_feedforward, [
                seg(y, 0, d, ldo=d, bias=pw.f(lay + ".linear2.bias"), res=self.xdec, ldres=d)]))
            # norm3, then the shared decoder.norm on its (rounded) output -> hs[li]: one launch
            ops.append(K.LayerNormChainOp(y, *n3))
        # ---- heads on all decoder layers at once
        rh = nl * rq
        hs2 = self.hs.view(rh, d)
        h1, h2 = z(rh, d), z(rh, d)
        self.delta = z(rh, 4)
        self.logits = z(rh, self.ldc)
        bb = "bbox_embed.layers"
        if chain_plain and self.ncls <= 1024:
            # class_embed and the bbox_embed MLP on the hidden states of all decoder layers: one launch
            ops.append(row_chain("heads", hs2, d, d, rh, [
                dict(kind="side", w="class_embed.weight", b="class_embed.bias", out=self.logits, ldo=self.ldc),
                dict(kind="full", w=bb + ".0.weight", b=bb + ".0.bias", relu=True),
                dict(kind="full", w=bb + ".1.weight", b=bb + ".1.bias", relu=True),
                dict(kind="side", w=bb + ".2.weight", b=bb + ".2.bias", out=self.delta, ldo=4)]))
        else:
            ops.append(GemmOp(hs2, pw.w(bb + ".0.weight"), rh, d, d, [seg(h1, 0, d, ldo=d, bias=pw.f(bb + ".0.bias"), act=ACT_RELU)]))
            ops.append(GemmOp(h1, pw.w(bb + ".1.weight"), rh, d, d, [seg(h2, 0, d, ldo=d, bias=pw.f(bb + ".1.bias"), act=ACT_RELU)]))
            ops.append(GemmOp(h2, pw.w(bb + ".2.weight"), rh, 4, d, [seg(self.delta, 0, 4, ldo=4, bias=pw.f(bb + ".2.bias"))]))
            ops.append(GemmOp(hs2, pw.w("class_embed.weight"), rh, self.ncls, d, [
                seg(self.logits, 0, self.ncls, ldo=self.ldc, bias=pw.f("class_embed.bias"))]))
        self.query_feat = pw.w("query_feat.weight", lambda w_: w_[:nq], "g0")
        self.refpoint = pw.f("refpoint_embed.weight", lambda w_: w_[:nq], "g0")
        # ---- fused glue launches (gather of the selected rows, decoder inputs, final boxes)
        code = K._nat.dtype_code(self.T)
        f32 = lambda *s_: self._own(torch.zeros(*s_, dtype=torch.float32, device=dev))
        self._pad_state = None
        props0, valid0 = self._proposals(None)
        self._props_static, self._valid_static = props0.contiguous(), valid0.reshape(-1).to(torch.uint8)
        self.props = self._props_static.clone()
        self.topk_idx = self._own(torch.zeros(B, nq, dtype=torch.int64, device=dev))
        self.enc_logits_sel, self.enc_boxes = z(B, nq, self.ncls), z(B, nq, 4)
        self.props_sel = f32(B, nq, 4)
        self.coord = z(nl, B, nq, 4)
        dim_t = torch.arange(d // 2, dtype=torch.float32, device=dev)
        self.dim_t = (10000 ** (2 * (dim_t // 2) / (d // 2))).contiguous()          # transformer.py:46-47
        ptr = lambda t_: t_.data_ptr()
        self.op_gather = K.RawOp("lwdetr_select_gather", (
            ptr(self.om), ptr(self.enc_cls), self.ldc_enc, ptr(self.props), ptr(self.topk_idx), ptr(self.om_sel),
            ptr(self.enc_logits_sel), ptr(self.props_sel), B, S, d, nq, self.ncls, code), keep=())
        # (with the chain kernel the class maxima are already written; the launch below then only serves forced selections' collect)
        self.op_rowmax = K.RawOp("lwdetr_rowmax", (ptr(self.enc_cls), self.ldc_enc, B * S, self.ncls, ptr(self.cls_max), code), keep=())
        self.op_topk = K.RawOp("lwdetr_topk", (ptr(self.cls_max), B, S, nq, ptr(self.topk_idx), None, 0), keep=())
        self.op_dec_inputs = K.RawOp("lwdetr_decoder_inputs", (
            ptr(self.enc_delta), ptr(self.props_sel), ptr(self.refpoint), ptr(self.vr), L, ptr(self.query_feat),
            ptr(self.dim_t), ptr(self.enc_boxes), ptr(self.ref), ptr(self.sine), ptr(self.xdec), B, nq, d, code), keep=())
        self.op_finalize = K.RawOp("lwdetr_finalize_outputs", (
            ptr(self.delta), ptr(self.ref), B * nq, ptr(self.coord), nl * B * nq, ptr(self.logits), self.ldc, self.ncls,
            ptr(self.logits), 0, code), keep=())     # args 3 (boxes), 8 (logits), 9 (rows per layer there): the call's output tensors

    # ------------------------------------------------------------------------------- per-call host-side glue
    def _masks(self, mask):
        """Per-level nearest-resized masks (backbone.py:155-158) -> (mask_flat (B,S) bool, valid ratios (B,L,2))."""
        flat, vr = [], []
        for (hl, wl) in self.level_hw:
            m = F.interpolate(mask[None].float(), size=(hl, wl)).to(torch.bool)[0]
            flat.append(m.flatten(1))
            vr.append(torch.stack([(~m[:, 0, :]).sum(1).float() / wl, (~m[:, :, 0]).sum(1).float() / hl], -1))
        return torch.cat(flat, 1), torch.stack(vr, 1)

    def _proposals(self, mask_flat):
        """Anchor proposals + validity (transformer.py:71-125, unsigmoid=False). mask_flat None = no padding."""
        B, dev = self.B, self.dev
        props, cur = [], 0
        for lvl, (hl, wl) in enumerate(self.level_hw):
            gy, gx = torch.meshgrid(torch.arange(hl, dtype=torch.float32, device=dev),
                                    torch.arange(wl, dtype=torch.float32, device=dev), indexing="ij")
            grid = torch.stack([gx, gy], -1)[None].expand(B, -1, -1, -1)
            if mask_flat is None:
                scale = torch.tensor([wl, hl], dtype=torch.float32, device=dev).view(1, 1, 1, 2)
            else:
                m = mask_flat[:, cur:cur + hl * wl].view(B, hl, wl)
                scale = torch.stack([(~m[:, 0, :]).sum(1), (~m[:, :, 0]).sum(1)], 1).view(B, 1, 1, 2).float()
            grid = (grid + 0.5) / scale
            wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
            props.append(torch.cat([grid, wh], -1).view(B, -1, 4))
            cur += hl * wl
        props = torch.cat(props, 1)
        valid = ((props > 0.01) & (props < 0.99)).all(-1)
        if mask_flat is not None:
            valid = valid & ~mask_flat
        props = props * valid.unsqueeze(-1)
        return props, valid

    @torch.no_grad()
    def _set_padding_state(self, mask):
        """Per-call masks -> row masks, valid ratios and anchor proposals; the no-padding state is set once and reused."""
        has_pad = mask is not None and bool(mask.any())
        if not has_pad:
            if self._pad_state != "nopad":
                self.props.copy_(self._props_static)
                self.rowvalid.copy_(self._valid_static)
                self.notpad.fill_(1)
                self.vr.fill_(1.0)
                self._pad_state = "nopad"
            return
        mask_flat, vr = self._masks(mask)
        props, valid = self._proposals(mask_flat)
        self.props.copy_(props)
        self.rowvalid.copy_(valid.reshape(-1).to(torch.uint8))
        self.notpad.copy_((~mask_flat).reshape(-1).to(torch.uint8))
        self.vr.copy_(vr)
        self._pad_state = "pad"

    @torch.no_grad()
    def run(self, images, mask=None, forced_topk=None, collect=None, into=None):
        """images (B,3,H,W) on the plan's device; mask (B,H,W) bool or None (= no padding). ``into`` = (outputs, b0): write this
        batch's results into images [b0, b0 + B) of preallocated full-batch tensors (``alloc_outputs``) instead of fresh ones -
        the launch chains of ``LWDETR._forward_chains`` fill one set of output tensors, no concatenation afterwards."""
        B, S, d, nq, T = self.B, self.S, self.d, self.nq, self.T
        stream = K._nat.stream_ptr(self.dev)
        if images.dtype == T and images.is_contiguous() and images.device == self.x.device and images.data_ptr() % 16 == 0:
            self._img_ref = images                                   # the patch GEMM reads the caller's tensor in place
        else:
            if self.images is None:
                self.images = self._z(B, 3, self.H, self.W)
            self.images.copy_(images)
            self._img_ref = self.images
        if self.stem_op is not None:
            self.stem_op.set_image(self._img_ref.data_ptr())
        else:
            self.patch_op.desc.A = self._img_ref.data_ptr()
        self._set_padding_state(mask)
        for op in self.ops_backbone:
            op(stream)
        for op in self.ops_enc:
            op(stream)
        # ---- two-stage selection (group 0 only at inference, transformer.py:229-264)
        if forced_topk is None:
            if not self.use_chain:
                self.op_rowmax(stream)
            self.op_topk(stream)
        else:
            self.topk_idx.copy_(forced_topk)
        # user-visible outputs are fresh tensors of this call, written directly by the kernels that produce them (no clone /
        # slice-copy launches): encoder logits by the gather, encoder boxes by the decoder-input kernel, the decoder layers'
        # boxes and the contiguous copy of their class logits (the GEMM output rows are padded to ldc) by one final launch
        nl, ncls = self.cfg.dec_layers, self.ncls
        if into is None:
            (enc_logits, enc_boxes, cls, coord), b0 = self.alloc_outputs(B), 0
        else:
            (enc_logits, enc_boxes, cls, coord), b0 = into
        total, isz = cls.shape[1], cls.element_size()
        self.op_gather.call_with(stream, {6: enc_logits.data_ptr() + b0 * nq * ncls * isz})
        for op in self.ops_sel:
            op(stream)
        self.op_dec_inputs.call_with(stream, {7: enc_boxes.data_ptr() + b0 * nq * 4 * isz})
        for op in self.ops_dec:
            op(stream)
        self.op_finalize.call_with(stream, {3: coord.data_ptr() + b0 * nq * 4 * isz, 8: cls.data_ptr() + b0 * nq * ncls * isz,
                                            9: total * nq})
        if into is not None:
            return None
        out = self.output_dict(enc_logits, enc_boxes, cls, coord)
        if collect is not None:
            if forced_topk is not None and not self.use_chain:
                self.op_rowmax(stream)
            collect.update({"topk_idx": self.topk_idx.clone(), "enc.class_max": self.cls_max.clone(), "memory": self.memory.view(B, S, d).clone(),
                            "taps_cat": self.taps_cat.clone(), "x": self.x.clone(), "hs": self.hs.clone(),
                            "om": self.om.view(B, S, d).clone(), "launch_chains": 1})
        return out

    def alloc_outputs(self, total):
        """Fresh user-visible tensors for ``total`` images: encoder logits / boxes, all decoder layers' logits / boxes."""
        nl, nq, ncls = self.cfg.dec_layers, self.nq, self.ncls
        empty = lambda *sh: torch.empty(*sh, dtype=self.T, device=self.dev)
        return empty(total, nq, ncls), empty(total, nq, 4), empty(nl, total, nq, ncls), empty(nl, total, nq, 4)

    def output_dict(self, enc_logits, enc_boxes, cls, coord):
        out = {"pred_logits": cls[-1], "pred_boxes": coord[-1]}
        if self.cfg.aux_loss:
            out["aux_outputs"] = [{"pred_logits": a, "pred_boxes": b} for a, b in zip(cls[:-1], coord[:-1])]
        out["enc_outputs"] = {"pred_logits": enc_logits, "pred_boxes": enc_boxes}
        return out


class GraphedForward:
    """One ForwardPlan captured into a HIP graph (see ``LWDETR.capture``). All kernels of the plan are launched on the
    stream torch is capturing (``_native.stream_ptr``), so the graph holds the whole forward: ~130 launches at the cost
    of one ``hipGraphLaunch``."""

    def __init__(self, plan, images, postprocess=None, target_sizes=None):
        self.plan, dev = plan, plan.dev
        self.static_in = images.to(device=dev, dtype=plan.T).contiguous().clone()
        self.sizes = None if target_sizes is None else target_sizes.to(device=dev, dtype=torch.float32).contiguous().clone()
        self.post = postprocess

        def body():
            out = plan.run(self.static_in)
            if self.post is not None:
                return out, self.post.select(out["pred_logits"], out["pred_boxes"], self.sizes)
            return out, None

        with torch.cuda.device(dev):
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                       # warm-up off the capture: one-time setup, allocator
                body()
                body()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out, self.det = body()

    @torch.no_grad()
    def __call__(self, images=None):
        if images is not None and images.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(images)
        self.graph.replay()
        return self.out if self.det is None else (self.out, self.det)
