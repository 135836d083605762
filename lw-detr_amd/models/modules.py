"""Parameter containers with the reference's module tree, so ``state_dict()`` / ``load_state_dict(strict=True)`` are
key- and shape-compatible with reference checkpoints (key layout: SURVEY.md section 9.3).

These modules own parameters only; none of them computes - the forward pass is the launch plan of
``lwdetr_amd.engine`` (hand-written HIP kernels). Initial values follow the reference's initialisers
(cited per class) so a freshly built model starts in the same state distributionally.
"""
import copy
import math

import torch
from torch import nn

from ..configs import LEVEL_SCALE, VIT_SIZES
from ..ops.modules import MSDeformAttn



class _NoCompute(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is a parameter container; call LWDETR.forward (HIP engine)")


# ------------------------------------------------------------------------------------------------- ViT encoder
class ViTAttention(_NoCompute):          # models/backbone/vit.py:86-118 (use_cae=True)
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.proj = nn.Linear(dim, dim)


class Mlp(_NoCompute):                   # timm.models.layers.Mlp as used by vit.py:184
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class ViTBlock(_NoCompute):              # models/backbone/vit.py:143-193
    def __init__(self, dim, num_heads, window):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = ViTAttention(dim, num_heads)
        self.drop_path = nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, dim * 4)
        self.window = window
        self.gamma_1 = nn.Parameter(0.1 * torch.ones(dim))
        self.gamma_2 = nn.Parameter(0.1 * torch.ones(dim))


class PatchEmbed(_NoCompute):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=16, stride=16)


class ViT(_NoCompute):                   # models/backbone/vit.py:225-341
    def __init__(self, embed_dim, depth, num_heads, window_block_indexes, out_feature_indexes):
        super().__init__()
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.patch_embed = PatchEmbed(embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, (224 // 16) ** 2 + 1, embed_dim))
        self.blocks = nn.ModuleList(ViTBlock(embed_dim, num_heads, i in window_block_indexes) for i in range(depth))
        self.window_block_indexes = list(window_block_indexes)
        idx = [i if i >= 0 else i + depth for i in out_feature_indexes]
        self.out_feature_indexes = [i for i in range(depth) if i in idx]
        assert self.out_feature_indexes[-1] == depth - 1
        self._out_feature_channels = [embed_dim] * len(self.out_feature_indexes)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)


# --------------------------------------------------------------------------------------------------- projector
class LayerNorm2d(_NoCompute):           # models/backbone/projector.py:21-47
    def __init__(self, c, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.eps = eps


class ConvX(_NoCompute):                 # models/backbone/projector.py:85-98
    def __init__(self, c_in, c_out, kernel=3, stride=1, act="relu"):
        super().__init__()
        self.conv = nn.Conv2d(c_in, c_out, kernel, stride, kernel // 2, bias=False)
        self.bn = nn.BatchNorm2d(c_out)
        self.kernel, self.stride, self.act_name = kernel, stride, act


class Bottleneck(_NoCompute):            # models/backbone/projector.py:101-114 (shortcut=False, e=1.0)
    def __init__(self, c):
        super().__init__()
        self.cv1 = ConvX(c, c, 3, 1, act="silu")
        self.cv2 = ConvX(c, c, 3, 1, act="silu")


class C2f(_NoCompute):                   # models/backbone/projector.py:117-132
    def __init__(self, c1, c2, n=3):
        super().__init__()
        self.c = int(c2 * 0.5)
        self.cv1 = ConvX(c1, 2 * self.c, 1, 1, act="silu")
        self.cv2 = ConvX((2 + n) * self.c, c2, 1, 1, act="silu")
        self.m = nn.ModuleList(Bottleneck(self.c) for _ in range(n))


class MultiScaleProjector(_NoCompute):   # models/backbone/projector.py:135-212
    def __init__(self, in_channels, out_channels, scale_factors):
        super().__init__()
        self.scale_factors = list(scale_factors)
        sampling, stages = [], []
        for scale in scale_factors:
            per_tap = []
            for c in in_channels:
                if scale == 2.0:
                    if c > 512:
                        layers = [ConvX(c, c // 2, kernel=1), nn.ConvTranspose2d(c // 2, c // 4, 2, 2)]
                        c_out = c // 4
                    else:
                        layers = [nn.ConvTranspose2d(c, c // 2, 2, 2)]
                        c_out = c // 2
                elif scale == 1.0:
                    layers, c_out = [], c
                elif scale == 0.5:
                    layers, c_out = [ConvX(c, c, 3, 2)], c
                else:
                    raise NotImplementedError(f"Unsupported scale_factor:{scale} (LW-DETR configs use P3/P4/P5)")
                per_tap.append(nn.Sequential(*layers))
            sampling.append(nn.ModuleList(per_tap))
            stages.append(nn.Sequential(C2f(c_out * len(in_channels), out_channels, 3), LayerNorm2d(out_channels)))
        self.stages_sampling = nn.ModuleList(sampling)
        self.stages = nn.ModuleList(stages)


class PositionEmbeddingSine(_NoCompute):
    """Kept for module-tree parity (``backbone.1``); it has no parameters and its output is never consumed by the
    decoder (reference ``models/transformer.py:466-517`` ignores ``pos``), so the engine does not compute it."""

    def __init__(self, num_pos_feats):
        super().__init__()
        self.num_pos_feats = num_pos_feats


class Backbone(_NoCompute):              # models/backbone/backbone.py:31-144 (ViT encoders)
    def __init__(self, name, depth, window_block_indexes, out_channels, out_feature_indexes, projector_scale):
        super().__init__()
        if name not in VIT_SIZES:
            raise NotImplementedError(f"Backbone {name} is not supported (LW-DETR configs use vit_tiny/small/base)")
        dim, heads = VIT_SIZES[name]
        assert window_block_indexes is not None and len(projector_scale) > 0
        assert sorted(projector_scale) == list(projector_scale), "projector scales must ascend (P3 < P4 < P5)"
        self.name = name
        self.encoder = ViT(dim, depth, heads, window_block_indexes, out_feature_indexes)
        self.projector_scale = list(projector_scale)
        self.projector = MultiScaleProjector(self.encoder._out_feature_channels, out_channels,
                                             [LEVEL_SCALE[l] for l in projector_scale])


class Joiner(nn.Sequential):             # models/backbone/__init__.py:11-35
    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)

    def forward(self, *a, **k):
        raise RuntimeError("Joiner is a parameter container; call LWDETR.forward (HIP engine)")


# ------------------------------------------------------------------------------------------------- transformer
class MLP(_NoCompute):                   # models/transformer.py:28-39
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))


class MultiheadAttention(_NoCompute):    # models/attention.py:58-130 (packed in-projection)
    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)


class TransformerDecoderLayer(_NoCompute):   # models/transformer.py:430-463
    def __init__(self, d_model, sa_nhead, ca_nhead, dim_feedforward, dropout, n_levels, n_points):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, sa_nhead)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.cross_attn = MSDeformAttn(d_model, n_levels=n_levels, n_heads=ca_nhead, n_points=n_points)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)


class TransformerDecoder(_NoCompute):    # models/transformer.py:291-312
    def __init__(self, layer, num_layers, d_model):
        super().__init__()
        self.layers = nn.ModuleList(copy.deepcopy(layer) for _ in range(num_layers))
        self.num_layers, self.d_model = num_layers, d_model
        self.norm = nn.LayerNorm(d_model)
        self.ref_point_head = MLP(2 * d_model, d_model, d_model, 2)
        self.bbox_embed = None


class Transformer(_NoCompute):           # models/transformer.py:128-187
    def __init__(self, d_model, sa_nhead, ca_nhead, num_queries, num_decoder_layers, dim_feedforward, dropout,
                 group_detr, two_stage, num_feature_levels, dec_n_points):
        super().__init__()
        layer = TransformerDecoderLayer(d_model, sa_nhead, ca_nhead, dim_feedforward, dropout, num_feature_levels,
                                        dec_n_points)
        self.decoder = TransformerDecoder(layer, num_decoder_layers, d_model)
        self.two_stage = two_stage
        if two_stage:
            self.enc_output = nn.ModuleList(nn.Linear(d_model, d_model) for _ in range(group_detr))
            self.enc_output_norm = nn.ModuleList(nn.LayerNorm(d_model) for _ in range(group_detr))
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        self.num_queries, self.d_model, self.dec_layers = num_queries, d_model, num_decoder_layers
        self.group_detr, self.num_feature_levels = group_detr, num_feature_levels
        self.sa_nhead, self.ca_nhead, self.dec_n_points = sa_nhead, ca_nhead, dec_n_points


def focal_prior_bias(num_classes, prior_prob=0.01):
    return torch.ones(num_classes) * (-math.log((1 - prior_prob) / prior_prob))   # models/lwdetr.py:85-87
