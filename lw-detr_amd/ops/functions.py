"""Multi-scale deformable attention operator - same signatures as the reference, HIP kernel underneath.

Reference: ``models/ops/functions/ms_deform_attn_func.py:26-49`` (MSDeformAttnFunction), ``:52-75``
(ms_deform_attn_core_pytorch) and the pybind module ``MultiScaleDeformableAttention``
(``models/ops/src/vision.cpp:13-16``, ``models/ops/src/ms_deform_attn.h:19-60``).

Differences, all deliberate: f16 / bf16 are accepted in addition to f32 / f64 (reference: ``ms_deform_attn_cuda.cu:64``
dispatches float/double only); ``im2col_step`` is validated like the reference (``ms_deform_attn_cuda.cu:50-52``) but the
whole batch is one launch; launch failures raise instead of being printed (``ms_deform_im2col_cuda.cuh:948-952``).
The backward (``ms_deform_attn_backward``, float32 / float64 like the reference) is implemented too, so the Function is
differentiable (``torch.autograd.gradcheck`` at the reference's own test shapes, ``tests/test_gpu_msda.py``).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _native as N


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """value (N,S,M,D), spatial_shapes (L,2) int64, level_start_index (L,) int64, sampling_loc (N,Lq,M,L,P,2),
    attn_weight (N,Lq,M,L,P) -> (N, Lq, M*D). All tensors contiguous, on one ROCm device."""
    tensors = (value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    names = ("value", "spatial_shapes", "level_start_index", "sampling_loc", "attn_weight")
    for t, n in zip(tensors, names):
        if not t.is_contiguous():
            raise RuntimeError(f"{n} tensor has to be contiguous")
        if not t.is_cuda:
            raise RuntimeError(f"{n} must be a CUDA tensor")          # reference: AT_ASSERTM(... is_cuda ...)
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes / level_start_index must be int64")
    if sampling_loc.dtype != value.dtype or attn_weight.dtype != value.dtype:
        raise RuntimeError("value, sampling_loc and attn_weight must share a dtype")
    b, s, m, d = value.shape
    l = spatial_shapes.shape[0]
    q, p = sampling_loc.shape[1], sampling_loc.shape[4]
    step = min(b, int(im2col_step)) if b > 0 else 1
    if step <= 0 or b % step != 0:
        raise RuntimeError(f"batch({b}) must divide im2col_step({step})")
    out = torch.empty(b, q, m * d, dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device):
        rc = N.lib().lwdetr_msda_forward(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                         sampling_loc.data_ptr(), attn_weight.data_ptr(), out.data_ptr(), b, s, m, d,
                                         l, q, p, N.dtype_code(value.dtype), N.stream_ptr(value.device))
    N.check(rc, "ms_deform_attn_forward")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step):
    """-> [grad_value (N,S,M,D), grad_sampling_loc (N,Lq,M,L,P,2), grad_attn_weight (N,Lq,M,L,P)]: the reference's
    ``ms_deform_attn_backward`` (``models/ops/src/ms_deform_attn.h:37-60``, ``cuda/ms_deform_attn_cuda.cu:83-153``).
    float32 / float64 only, as in the reference (``.cu:132``)."""
    tensors = (value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output)
    names = ("value", "spatial_shapes", "level_start_index", "sampling_loc", "attn_weight", "grad_output")
    for t, n in zip(tensors, names):
        if not t.is_contiguous():
            raise RuntimeError(f"{n} tensor has to be contiguous")
        if not t.is_cuda:
            raise RuntimeError(f"{n} must be a CUDA tensor")
    if value.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("ms_deform_attn_backward: float32 / float64 only (the reference dispatches the same two types)")
    if any(t.dtype != value.dtype for t in (sampling_loc, attn_weight, grad_output)):
        raise RuntimeError("value, sampling_loc, attn_weight and grad_output must share a dtype")
    b, s, m, d = value.shape
    l = spatial_shapes.shape[0]
    q, p = sampling_loc.shape[1], sampling_loc.shape[4]
    step = min(b, int(im2col_step)) if b > 0 else 1
    if step <= 0 or b % step != 0:
        raise RuntimeError(f"batch({b}) must divide im2col_step({step})")
    gv, gl, ga = torch.empty_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)
    with torch.cuda.device(value.device):
        rc = N.lib().lwdetr_msda_backward(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                          sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                                          gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), b, s, m, d, l, q, p,
                                          N.dtype_code(value.dtype), N.stream_ptr(value.device))
    N.check(rc, "ms_deform_attn_backward")
    return [gv, gl, ga]


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        out = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                     attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, aw = ctx.saved_tensors
        gv, gl, ga = ms_deform_attn_backward(value, shapes, lsi, loc, aw, grad_output.contiguous(), ctx.im2col_step)
        return gv, None, None, gl, ga, None


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """Signature of the reference's debug core (value is (N, M, D, S) here); runs the HIP kernel.

    attention_weights may be (N,Lq,M,L,P) or (N,Lq,M,L*P) as in ``ms_deform_attn.py:134``."""
    n, m, d, s = value.shape
    _, lq, _, l, p, _ = sampling_locations.shape
    shapes = torch.as_tensor(value_spatial_shapes, dtype=torch.int64, device=value.device).reshape(l, 2).contiguous()
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1])).contiguous()
    v = value.permute(0, 3, 1, 2).contiguous()
    aw = attention_weights.reshape(n, lq, m, l, p).contiguous()
    return ms_deform_attn_forward(v, shapes, lsi, sampling_locations.contiguous(), aw, 64 if n % 64 == 0 else n)
