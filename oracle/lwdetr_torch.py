"""TEST INFRASTRUCTURE ONLY - functional fp32/fp64 restatement of the LW-DETR inference forward.

This is the oracle for the HIP path: a state-dict driven, module-free restatement of the reference's
arithmetic, written from the reference's behaviour (citations below) and pinned against golden vectors that
were produced by running the *unmodified* reference (``oracle/gen_golden.py`` -> ``tests/golden``).
It must never be imported by the product package; see ``oracle/__init__.py``.

Reference map (file:line in /root/reference):
  vit()                models/backbone/vit.py:26-54 (abs pos), :79-83 (patch embed), :120-140 (attention),
                       :195-222 (block), :343-365 (window-major reorder, taps)
  projector()          models/backbone/projector.py:85-98 (ConvX), :101-132 (Bottleneck, C2f), :21-47 (LN2d),
                       :177-193 (resamplers), :214-241 (forward)
  masks/valid ratios   models/backbone/backbone.py:155-158, models/transformer.py:189-196
  proposals()          models/transformer.py:71-125
  two_stage()          models/transformer.py:224-264
  decoder()            models/transformer.py:328-427 (get_reference, layers), :42-68 (sine embed), :466-517 (layer)
  mha()                models/attention.py:215-451, :507-606
  msda()               models/ops/modules/ms_deform_attn.py:96-144, models/ops/functions/ms_deform_attn_func.py:52-75
  heads()              models/lwdetr.py:149-173
  postprocess()        models/lwdetr.py:515-544, util/box_ops.py:21-25
"""
import math

import torch
import torch.nn.functional as F

_VIT = {"vit_tiny": (192, 12), "vit_small": (384, 12), "vit_base": (768, 12)}   # backbone.py:46-51
_LEVEL_SCALE = {"P3": 2.0, "P4": 1.0, "P5": 0.5}                                 # backbone.py:124-129


def _lin(sd, key, x):
    return F.linear(x, sd[key + ".weight"], sd.get(key + ".bias"))


def _ln(sd, key, x, eps):
    return F.layer_norm(x, x.shape[-1:], sd[key + ".weight"], sd[key + ".bias"], eps)


def _mlp(sd, key, x, n):
    for i in range(n):
        x = _lin(sd, f"{key}.layers.{i}", x)
        if i < n - 1:
            x = F.relu(x)
    return x


# ----------------------------------------------------------------------------------------------- ViT
def abs_pos(pos_embed, hp, wp):
    """(1, 1+14*14, C) -> (1, hp, wp, C): drop cls token, bicubic resize, align_corners=False."""
    p = pos_embed[:, 1:]
    s = int(math.isqrt(p.shape[1]))
    if (s, s) != (hp, wp):
        p = F.interpolate(p.reshape(1, s, s, -1).permute(0, 3, 1, 2), size=(hp, wp), mode="bicubic",
                          align_corners=False).permute(0, 2, 3, 1)
        return p
    return p.reshape(1, hp, wp, -1)


def vit_attention(sd, pre, x, heads):
    n_seq, n_tok, c = x.shape
    hd = c // heads
    bias = torch.cat([sd[pre + ".q_bias"], torch.zeros_like(sd[pre + ".v_bias"]), sd[pre + ".v_bias"]])
    qkv = F.linear(x, sd[pre + ".qkv.weight"], bias).reshape(n_seq, n_tok, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(-1)
    out = (att @ v).transpose(1, 2).reshape(n_seq, n_tok, c)
    return _lin(sd, pre + ".proj", out)


def vit(sd, cfg, images, collect=None):
    """images (B,3,H,W) -> list of tap feature maps (B, C, H/16, W/16)."""
    c, heads = _VIT[cfg.encoder]
    pre = "backbone.0.encoder"
    x = F.conv2d(images, sd[pre + ".patch_embed.proj.weight"], sd[pre + ".patch_embed.proj.bias"], stride=16)
    x = x.permute(0, 2, 3, 1)
    b, hp, wp, _ = x.shape
    x = x + abs_pos(sd[pre + ".pos_embed"], hp, wp)
    assert hp % 4 == 0 and wp % 4 == 0
    h, w = hp // 4, wp // 4
    x = x.reshape(b, 4, h, 4, w, c).permute(0, 1, 3, 2, 4, 5).reshape(b * 16, h * w, c)
    if collect is not None:
        collect["vit.embed"] = x.reshape(b, 16 * h * w, c)
    depth = cfg.vit_encoder_num_layers
    taps_at = sorted(i if i >= 0 else i + depth for i in cfg.out_feature_indexes)
    taps = []
    for i in range(depth):
        blk = f"{pre}.blocks.{i}"
        y = _ln(sd, blk + ".norm1", x, 1e-6)
        if i not in cfg.window_block_indexes:                      # global block: 16 windows -> one sequence
            y = y.reshape(b, 16 * h * w, c)
        y = vit_attention(sd, blk + ".attn", y, heads).reshape(b * 16, h * w, c)
        x = x + sd[blk + ".gamma_1"] * y
        y = _lin(sd, blk + ".mlp.fc2", F.gelu(_lin(sd, blk + ".mlp.fc1", _ln(sd, blk + ".norm2", x, 1e-6))))
        x = x + sd[blk + ".gamma_2"] * y
        if collect is not None:
            collect[f"vit.block{i}"] = x.reshape(b, 16 * h * w, c)
        if i in taps_at:
            taps.append(x.reshape(b, 4, 4, h, w, c).permute(0, 5, 1, 3, 2, 4).reshape(b, c, hp, wp))
    return taps


# ----------------------------------------------------------------------------------------- projector
def _convx(sd, key, x, k, stride, act):
    x = F.conv2d(x, sd[key + ".conv.weight"], None, stride=stride, padding=k // 2)
    x = F.batch_norm(x, sd[key + ".bn.running_mean"], sd[key + ".bn.running_var"], sd[key + ".bn.weight"],
                     sd[key + ".bn.bias"], training=False, eps=1e-5)
    return F.silu(x) if act == "silu" else F.relu(x)


def _ln2d(sd, key, x, eps=1e-6):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return sd[key + ".weight"][:, None, None] * x + sd[key + ".bias"][:, None, None]


def projector(sd, cfg, taps, collect=None):
    pre = "backbone.0.projector"
    c_in = taps[0].shape[1]
    levels = []
    for li, name in enumerate(cfg.projector_scale):
        scale = _LEVEL_SCALE[name]
        feats = []
        for ti, t in enumerate(taps):
            sp = f"{pre}.stages_sampling.{li}.{ti}"
            if scale == 2.0:
                if c_in > 512:
                    t = _convx(sd, sp + ".0", t, 1, 1, "relu")
                    t = F.conv_transpose2d(t, sd[sp + ".1.weight"], sd[sp + ".1.bias"], stride=2)
                else:
                    t = F.conv_transpose2d(t, sd[sp + ".0.weight"], sd[sp + ".0.bias"], stride=2)
            elif scale == 0.5:
                t = _convx(sd, sp + ".0", t, 3, 2, "relu")
            feats.append(t)
        x = torch.cat(feats, 1) if len(feats) > 1 else feats[0]
        st = f"{pre}.stages.{li}.0"                                  # C2f(c1, d, n=3, shortcut=False, e=0.5)
        y = _convx(sd, st + ".cv1", x, 1, 1, "silu")
        half = y.shape[1] // 2
        ys = [y[:, :half], y[:, half:]]
        for m in range(3):
            z = _convx(sd, f"{st}.m.{m}.cv1", ys[-1], 3, 1, "silu")
            ys.append(_convx(sd, f"{st}.m.{m}.cv2", z, 3, 1, "silu"))
        y = _convx(sd, st + ".cv2", torch.cat(ys, 1), 1, 1, "silu")
        y = _ln2d(sd, f"{pre}.stages.{li}.1", y)
        if collect is not None:
            collect[f"proj.level{li}"] = y
        levels.append(y)
    return levels


# -------------------------------------------------------------------------------------- transformer
def sine_embed(pos, dim):
    """(B, nq, 4) -> (B, nq, 4*dim), order (y, x, w, h), temperature 10000, scale 2*pi."""
    dim_t = torch.arange(dim, dtype=torch.float32, device=pos.device)
    dim_t = (10000 ** (2 * (dim_t // 2) / dim)).to(pos.dtype)
    out = []
    for idx in (1, 0, 2, 3):
        e = pos[:, :, idx, None] * (2 * math.pi) / dim_t
        out.append(torch.stack((e[:, :, 0::2].sin(), e[:, :, 1::2].cos()), dim=3).flatten(2))
    return torch.cat(out, dim=2)


def proposals(memory, mask_flat, shapes):
    """Anchor proposals (unsigmoid=False branch) + memory rows zeroed where the proposal is invalid."""
    b = memory.shape[0]
    props, cur = [], 0
    for lvl, (hl, wl) in enumerate(shapes):
        m = mask_flat[:, cur:cur + hl * wl].view(b, hl, wl)
        valid_h = (~m[:, :, 0]).sum(1)
        valid_w = (~m[:, 0, :]).sum(1)
        gy, gx = torch.meshgrid(torch.arange(hl, dtype=torch.float32, device=memory.device),
                                torch.arange(wl, dtype=torch.float32, device=memory.device), indexing="ij")
        grid = torch.stack([gx, gy], -1)[None].expand(b, -1, -1, -1)
        scale = torch.stack([valid_w, valid_h], 1).view(b, 1, 1, 2)
        grid = (grid + 0.5) / scale
        wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
        props.append(torch.cat([grid, wh], -1).view(b, -1, 4))
        cur += hl * wl
    props = torch.cat(props, 1)
    valid = ((props > 0.01) & (props < 0.99)).all(-1, keepdim=True)
    props = props.masked_fill(mask_flat.unsqueeze(-1), 0.0).masked_fill(~valid, 0.0)
    out_mem = memory.masked_fill(mask_flat.unsqueeze(-1), 0.0).masked_fill(~valid, 0.0)
    return out_mem.to(memory.dtype), props.to(memory.dtype)


def reparam(delta, ref):
    return torch.cat([delta[..., :2] * ref[..., 2:] + ref[..., :2], delta[..., 2:].exp() * ref[..., 2:]], -1)


def mha(sd, pre, q_in, k_in, v_in, heads):
    d = q_in.shape[-1]
    w, bias = sd[pre + ".in_proj_weight"], sd[pre + ".in_proj_bias"]
    q = F.linear(q_in, w[:d], bias[:d])
    k = F.linear(k_in, w[d:2 * d], bias[d:2 * d])
    v = F.linear(v_in, w[2 * d:], bias[2 * d:])
    b, n, _ = q.shape
    hd = d // heads
    sp = lambda t: t.reshape(b, n, heads, hd).transpose(1, 2)
    att = ((sp(q) / math.sqrt(hd)) @ sp(k).transpose(-2, -1)).softmax(-1)
    out = (att @ sp(v)).transpose(1, 2).reshape(b, n, d)
    return _lin(sd, pre + ".out_proj", out)


def msda_core(value, shapes, loc, aw):
    """value (B,S,M,D); loc (B,Q,M,L,P,2) (x,y); aw (B,Q,M,L,P) -> (B,Q,M*D). grid_sample formulation."""
    b, _, m, d = value.shape
    q = loc.shape[1]
    out = value.new_zeros(b, q, m, d)
    cur = 0
    for lvl, (hl, wl) in enumerate(shapes):
        v = value[:, cur:cur + hl * wl].permute(0, 2, 3, 1).reshape(b * m, d, hl, wl)
        grid = (2 * loc[:, :, :, lvl] - 1).transpose(1, 2).flatten(0, 1)            # (B*M, Q, P, 2)
        samp = F.grid_sample(v, grid, mode="bilinear", padding_mode="zeros", align_corners=False)  # (B*M,D,Q,P)
        w = aw[:, :, :, lvl].transpose(1, 2).reshape(b * m, 1, q, -1)
        out += (samp * w).sum(-1).view(b, m, d, q).permute(0, 3, 1, 2)
        cur += hl * wl
    return out.reshape(b, q, m * d)


def msda(sd, pre, query, ref, memory, mask_flat, shapes, n_heads, n_points):
    b, q, d = query.shape
    n_levels = len(shapes)
    value = _lin(sd, pre + ".value_proj", memory).masked_fill(mask_flat[..., None], 0.0)
    off = _lin(sd, pre + ".sampling_offsets", query).view(b, q, n_heads, n_levels, n_points, 2)
    aw = _lin(sd, pre + ".attention_weights", query).view(b, q, n_heads, n_levels * n_points).softmax(-1)
    loc = ref[:, :, None, :, None, :2] + off / n_points * ref[:, :, None, :, None, 2:] * 0.5
    out = msda_core(value.view(b, -1, n_heads, d // n_heads), shapes, loc,
                    aw.view(b, q, n_heads, n_levels, n_points))
    return _lin(sd, pre + ".output_proj", out)


def forward(sd, cfg, images, mask=None, forced_topk=None, collect=None):
    """Full inference forward. Returns the reference's output dict (+ 'topk_idx' for teacher forcing)."""
    b, _, hh, ww = images.shape
    if mask is None:
        mask = torch.zeros(b, hh, ww, dtype=torch.bool, device=images.device)
    d = cfg.hidden_dim
    nq = cfg.num_queries
    levels = projector(sd, cfg, vit(sd, cfg, images, collect), collect)
    shapes = [tuple(l.shape[-2:]) for l in levels]
    masks = [F.interpolate(mask[None].float(), size=s).to(torch.bool)[0] for s in shapes]
    memory = torch.cat([l.flatten(2).transpose(1, 2) for l in levels], 1)
    mask_flat = torch.cat([m.flatten(1) for m in masks], 1)
    vr = torch.stack([torch.stack([(~m[:, 0, :]).sum(1).float() / m.shape[2],
                                   (~m[:, :, 0]).sum(1).float() / m.shape[1]], -1) for m in masks], 1).to(memory.dtype)
    t = "transformer"
    # two-stage query selection (group 0 only at inference)
    out_mem, props = proposals(memory, mask_flat, shapes)
    out_mem = _ln(sd, f"{t}.enc_output_norm.0", _lin(sd, f"{t}.enc_output.0", out_mem), 1e-5)
    enc_cls = _lin(sd, f"{t}.enc_out_class_embed.0", out_mem)
    enc_box = reparam(_mlp(sd, f"{t}.enc_out_bbox_embed.0", out_mem, 3), props)
    topk = torch.topk(enc_cls.max(-1)[0], nq, dim=1)[1] if forced_topk is None else forced_topk
    ref_ts = torch.gather(enc_box, 1, topk.unsqueeze(-1).expand(-1, -1, 4))
    mem_ts = torch.gather(out_mem, 1, topk.unsqueeze(-1).expand(-1, -1, d))
    if collect is not None:
        collect.update({"memory": memory, "enc.class_max": enc_cls.max(-1)[0], "enc.ref_ts": ref_ts})
    # decoder
    tgt = sd["query_feat.weight"][:nq].unsqueeze(0).expand(b, -1, -1)
    ref = reparam(sd["refpoint_embed.weight"][:nq].unsqueeze(0).expand(b, -1, -1), ref_ts)
    ref_in = ref[:, :, None] * torch.cat([vr, vr], -1)[:, None]                      # (B, nq, L, 4)
    qpos = _mlp(sd, f"{t}.decoder.ref_point_head", sine_embed(ref_in[:, :, 0], d // 2), 2)
    hs = []
    x = tgt
    for li in range(cfg.dec_layers):
        lay = f"{t}.decoder.layers.{li}"
        qk = x + qpos
        x = _ln(sd, lay + ".norm1", x + mha(sd, lay + ".self_attn", qk, qk, x, cfg.sa_nheads), 1e-5)
        x = _ln(sd, lay + ".norm2", x + msda(sd, lay + ".cross_attn", x + qpos, ref_in, memory, mask_flat, shapes,
                                              cfg.ca_nheads, cfg.dec_n_points), 1e-5)
        x = _ln(sd, lay + ".norm3", x + _lin(sd, lay + ".linear2", F.relu(_lin(sd, lay + ".linear1", x))), 1e-5)
        hs.append(_ln(sd, f"{t}.decoder.norm", x, 1e-5))
        if collect is not None:
            collect[f"dec.layer{li}"] = hs[-1]
    hs = torch.stack(hs)
    coord = reparam(_mlp(sd, "bbox_embed", hs, 3), ref[None])
    cls = _lin(sd, "class_embed", hs)
    out = {"pred_logits": cls[-1], "pred_boxes": coord[-1],
           "aux_outputs": [{"pred_logits": a, "pred_boxes": c} for a, c in zip(cls[:-1], coord[:-1])],
           "enc_outputs": {"pred_logits": _lin(sd, f"{t}.enc_out_class_embed.0", mem_ts), "pred_boxes": ref_ts},
           "topk_idx": topk}
    return out


def postprocess(outputs, target_sizes, num_select):
    logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
    prob = logits.sigmoid()
    scores, idx = torch.topk(prob.view(logits.shape[0], -1), num_select, dim=1)
    box_idx, labels = idx // logits.shape[2], idx % logits.shape[2]
    cx, cy, w, h = boxes.unbind(-1)
    w, h = w.clamp(min=0), h.clamp(min=0)
    xyxy = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)
    xyxy = torch.gather(xyxy, 1, box_idx.unsqueeze(-1).expand(-1, -1, 4))
    img_h, img_w = target_sizes.unbind(1)
    xyxy = xyxy * torch.stack([img_w, img_h, img_w, img_h], 1)[:, None, :]
    return [{"scores": s, "labels": l, "boxes": bx} for s, l, bx in zip(scores, labels, xyxy)]
