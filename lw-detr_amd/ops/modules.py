"""``MSDeformAttn`` module with the reference's parameters / state-dict keys and forward contract
(``models/ops/modules/ms_deform_attn.py:38-144``); projections run on the fused MFMA GEMM, the core on the fused
deformable-attention kernel (softmax + location arithmetic in the kernel prologue)."""
import math

import torch
from torch import nn

from .. import kernels as K


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError(f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()
        self._export = False

    def export(self):
        self._export = True

    def _reset_parameters(self):
        """Same initial state as the reference (``ms_deform_attn.py:79-94``): directional offset bias, zero weights."""
        nn.init.constant_(self.sampling_offsets.weight, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2).repeat(
            1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.view(-1))
        nn.init.constant_(self.attention_weights.weight, 0.0)
        nn.init.constant_(self.attention_weights.bias, 0.0)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.constant_(self.value_proj.bias, 0.0)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.0)

    # (L, P) pairs and head widths the fused softmax + sampling kernel is instantiated for (csrc/msda.hip); any other
    # configuration - including this module's defaults n_levels = n_points = 4 - runs softmax in torch and the generic
    # operator (lwdetr_msda_forward).
    _FUSED_LP = {(1, 2), (2, 4), (1, 4), (2, 2)}

    @torch.no_grad()      # INFERENCE-ONLY module: its Linear layers run on the fused GEMM kernel, which has no autograd
    # node; the differentiable piece of this package is the operator, MSDeformAttnFunction (forward + backward)
    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        n, lq, _ = query.shape
        _, s, _ = input_flatten.shape
        assert int((input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum()) == s
        if reference_points.shape[-1] != 4:
            raise ValueError("lwdetr_amd MSDeformAttn implements the 4-coordinate (box) reference form used by "
                             "LW-DETR (ms_deform_attn.py:125-127)")
        dt, dev = query.dtype, query.device
        d, m = self.d_model, self.n_heads
        lp = self.n_levels * self.n_points
        value = K.linear(input_flatten.reshape(n * s, d).contiguous(), self.value_proj.weight.to(dt),
                         self.value_proj.bias)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask.reshape(-1, 1), 0.0)
        w_oa = torch.cat([self.sampling_offsets.weight, self.attention_weights.weight], 0).to(dt).contiguous()
        b_oa = torch.cat([self.sampling_offsets.bias, self.attention_weights.bias], 0)
        oa = K.linear(query.reshape(n * lq, d).contiguous(), w_oa, b_oa)
        out = torch.empty(n * lq, d, dtype=dt, device=dev)
        ref = reference_points.float().contiguous()            # (N, Lq, L, 4): already scaled per level
        # the fused kernel takes unscaled boxes + per-level valid ratios; feed ratio 1 and level-0 boxes when all
        # levels carry the same box, otherwise fold the per-level boxes through the generic op
        fused = (self.n_levels, self.n_points) in self._FUSED_LP and (d // m) % 8 == 0 and dt != torch.float64
        fused = fused and bool((ref == ref[:, :, :1]).all())
        if not fused:
            off = oa[:, :m * lp * 2].reshape(n, lq, m, self.n_levels, self.n_points, 2).float()
            aw = oa[:, m * lp * 2:].reshape(n, lq, m, lp).float().softmax(-1)
            loc = ref[:, :, None, :, None, :2] + off / self.n_points * ref[:, :, None, :, None, 2:] * 0.5
            from .functions import ms_deform_attn_forward
            core = ms_deform_attn_forward(value.view(n, s, m, d // m), input_spatial_shapes.contiguous(),
                                          input_level_start_index.contiguous(), loc.to(dt).contiguous(),
                                          aw.view(n, lq, m, self.n_levels, self.n_points).to(dt).contiguous(),
                                          self.im2col_step if n % self.im2col_step == 0 else n)
            out = core.reshape(n * lq, d)
        else:
            vr = torch.ones(n, self.n_levels, 2, dtype=torch.float32, device=dev)
            K.MsdaFusedOp(value, input_spatial_shapes.contiguous(), input_level_start_index.contiguous(), oa,
                          oa.shape[1], m * lp * 2, ref[:, :, 0].contiguous(), vr, out, B=n, S=s, M=m, D=d // m,
                          L=self.n_levels, Q=lq, P=self.n_points)()
        y = K.linear(out, self.output_proj.weight.to(dt), self.output_proj.bias)
        return y.view(n, lq, d)
