"""GPU: the HIP deformable-attention op behind the reference's operator API, against the reference's KAT vectors
(models/ops/test.py), the C restatement oracle and size-independent properties."""
import os

import numpy as np
import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lsi(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1])).contiguous()


def _run(value, shapes, loc, aw, step=64):
    from lwdetr_amd.ops import MSDeformAttnFunction
    shapes = torch.as_tensor(shapes, dtype=torch.int64, device=DEV)
    b = value.shape[0]
    return MSDeformAttnFunction.apply(value.to(DEV).contiguous(), shapes, _lsi(shapes), loc.to(DEV).contiguous(),
                                      aw.to(DEV).contiguous(), step if b % step == 0 else b)


def test_reference_kat_double_and_float():
    g = load_golden("msda_op_kat")
    out = _run(torch.from_numpy(g["double_value"]), g["shapes"], torch.from_numpy(g["double_loc"]),
               torch.from_numpy(g["double_aw"]), step=2)
    assert torch.allclose(out.cpu(), torch.from_numpy(g["double_out"]))          # models/ops/test.py:56
    out = _run(torch.from_numpy(g["float_value"]), g["shapes"], torch.from_numpy(g["float_loc"]),
               torch.from_numpy(g["float_aw"]), step=2)
    assert torch.allclose(out.cpu(), torch.from_numpy(g["float_out"]), rtol=1e-2, atol=1e-3)   # test.py:82
    assert (out.cpu() - torch.from_numpy(g["float_out"])).abs().max() < 1e-7


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-6), (torch.float16, 2e-2), (torch.bfloat16, 1e-1)])
def test_out_of_bounds_golden(dtype, tol):
    g = load_golden("msda_op_kat")
    out = _run(torch.from_numpy(g["oob_value"]).to(dtype), g["oob_shapes"], torch.from_numpy(g["oob_loc"]).to(dtype),
               torch.from_numpy(g["oob_aw"]).to(dtype))
    assert (out.float().cpu() - torch.from_numpy(g["oob_out"])).abs().max().item() < tol


@pytest.mark.parametrize("cfg", [dict(B=8, shapes=[(40, 40)], M=16, D=16, Q=300, P=2),
                                 dict(B=3, shapes=[(80, 80), (20, 20)], M=24, D=16, Q=300, P=4),
                                 dict(B=2, shapes=[(7, 5), (3, 9), (4, 4)], M=2, D=8, Q=33, P=3),
                                 dict(B=2, shapes=[(9, 6)], M=3, D=5, Q=17, P=1)])
def test_against_c_oracle(cfg):
    from oracle import msda_c
    rng = np.random.default_rng(0)
    shapes = np.array(cfg["shapes"], dtype=np.int64)
    s = int((shapes[:, 0] * shapes[:, 1]).sum())
    b, m, d, q, p, l = cfg["B"], cfg["M"], cfg["D"], cfg["Q"], cfg["P"], len(cfg["shapes"])
    value = rng.standard_normal((b, s, m, d)).astype(np.float32)
    loc = (rng.random((b, q, m, l, p, 2)) * 1.3 - 0.15).astype(np.float32)
    loc[0, 0, 0, 0, 0] = [(2 + 0.5) / shapes[0, 1], (1 + 0.5) / shapes[0, 0]]     # exactly on a pixel centre
    aw = rng.random((b, q, m, l, p)).astype(np.float32)
    aw /= aw.sum((-1, -2), keepdims=True)
    ref = msda_c.msda_forward(value, shapes, loc, aw)
    out = _run(torch.from_numpy(value), shapes, torch.from_numpy(loc), torch.from_numpy(aw))
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-5


def test_properties_full_size_and_contract():
    """BASELINE-size checks that need no oracle: linearity in value and in the weights, zero weights -> zero,
    and the argument contract of the reference host wrapper (ms_deform_attn_cuda.cu:28-52)."""
    from lwdetr_amd.ops import MSDeformAttnFunction
    torch.manual_seed(0)
    b, shapes, m, d, q, p = 32, [(80, 80), (20, 20)], 24, 16, 300, 4
    sh = torch.tensor(shapes, dtype=torch.int64, device=DEV)
    s = 6800
    v1, v2 = torch.randn(b, s, m, d, device=DEV), torch.randn(b, s, m, d, device=DEV)
    loc = torch.rand(b, q, m, 2, p, 2, device=DEV) * 1.2 - 0.1
    aw = torch.rand(b, q, m, 2, p, device=DEV)
    f = lambda v, a: MSDeformAttnFunction.apply(v, sh, _lsi(sh), loc, a, 64 if b % 64 == 0 else b)
    o1, o2, o12 = f(v1, aw), f(v2, aw), f(v1 + 2 * v2, aw)
    assert (o12 - (o1 + 2 * o2)).abs().max().item() < 2e-4
    assert (f(v1, aw * 0.5) - 0.5 * o1).abs().max().item() < 1e-5
    assert f(v1, torch.zeros_like(aw)).abs().max().item() == 0.0
    assert o1.shape == (b, q, m * d)
    with pytest.raises(RuntimeError):
        MSDeformAttnFunction.apply(v1.transpose(1, 2), sh, _lsi(sh), loc, aw, 32)       # non-contiguous
    with pytest.raises(RuntimeError):
        MSDeformAttnFunction.apply(v1.cpu(), sh, _lsi(sh), loc, aw, 32)                 # wrong device
    with pytest.raises(RuntimeError):
        MSDeformAttnFunction.apply(v1[:30].contiguous(), sh, _lsi(sh), loc[:30].contiguous(), aw[:30].contiguous(), 7)
    empty = MSDeformAttnFunction.apply(v1, sh, _lsi(sh), loc[:, :0].contiguous(), aw[:, :0].contiguous(), 32)
    assert empty.shape == (b, 0, m * d)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float16, 2e-2)])
@pytest.mark.parametrize("lp", [(1, 2), (2, 4), (4, 4), (3, 1)])
def test_fused_module_matches_oracle_module(dtype, tol, lp):
    """MSDeformAttn module vs the torch restatement of ms_deform_attn.py:96-144: the (L, P) pairs of the five models on
    the fused prologue kernel, the module's own defaults (4, 4) and an odd pair on the generic operator."""
    from lwdetr_amd.ops import MSDeformAttn
    from lwdetr_amd.synth import synth_state_dict
    from oracle import lwdetr_torch as O
    l, p = lp
    d, m, b, q = 256, 16, 2, 50
    shapes = [(20, 24), (10, 12), (5, 6), (3, 3)][:l]
    s = sum(h * w for h, w in shapes)
    mod = MSDeformAttn(d, l, m, p)
    sd = synth_state_dict(mod.state_dict(), seed=5)
    mod.load_state_dict(sd)
    g = torch.Generator().manual_seed(1)
    query, memory = torch.randn(b, q, d, generator=g), torch.randn(b, s, d, generator=g)
    ref_box = torch.rand(b, q, 4, generator=g) * torch.tensor([1, 1, 0.3, 0.3]) + torch.tensor([0, 0, 0.02, 0.02])
    ref_in = ref_box[:, :, None].expand(-1, -1, l, -1).contiguous()
    mask = torch.zeros(b, s, dtype=torch.bool)
    mask[1, -17:] = True
    exp = O.msda({"x." + k: v for k, v in sd.items()}, "x", query, ref_in, memory, mask, shapes, m, p)
    sh = torch.tensor(shapes, dtype=torch.int64, device=DEV)
    mod = mod.to(DEV).to(dtype)
    out = mod(query.to(DEV, dtype), ref_in.to(DEV, dtype), memory.to(DEV, dtype), sh, _lsi(sh), mask.to(DEV))
    assert (out.float().cpu() - exp).abs().max().item() < tol * max(1.0, exp.abs().max().item())


def test_core_pytorch_signature_and_native_module_name():
    """a18 / B4: `ms_deform_attn_core_pytorch(value (N,M,D,S), shapes, loc, weights (N,Lq,M,L*P))` - the reference's debug
    core signature (models/ops/functions/ms_deform_attn_func.py:52-75) - and the native module imported BY ITS REFERENCE
    NAME from lw-detr_amd/compat (models/ops/src/vision.cpp:13-16: ms_deform_attn_forward / _backward, positional
    arguments as the pybind functions take them), both against the reference's known-answer vectors."""
    import importlib
    import sys
    from helpers import ROOT
    from lwdetr_amd.ops import ms_deform_attn_core_pytorch
    sys.path.insert(0, os.path.join(ROOT, "lw-detr_amd", "compat"))
    try:
        MSDA = importlib.import_module("MultiScaleDeformableAttention")
    finally:
        sys.path.pop(0)
    assert MSDA.__file__.endswith(os.path.join("compat", "MultiScaleDeformableAttention.py"))
    g = load_golden("msda_op_kat")
    v = torch.from_numpy(g["oob_value"]).to(DEV)
    sh = torch.from_numpy(g["oob_shapes"]).to(DEV)
    loc = torch.from_numpy(g["oob_loc"]).to(DEV)
    aw = torch.from_numpy(g["oob_aw"]).to(DEV)
    exp = torch.from_numpy(g["oob_out"]).to(DEV)
    n, s_, m_, d_ = v.shape
    out_mod = MSDA.ms_deform_attn_forward(v, sh, _lsi(sh), loc, aw, 2)
    assert (out_mod - exp).abs().max().item() < 1e-5
    lq, l_, p_ = loc.shape[1], loc.shape[3], loc.shape[4]
    out_core = ms_deform_attn_core_pytorch(v.permute(0, 2, 3, 1).contiguous(), [tuple(x) for x in sh.tolist()], loc,
                                           aw.reshape(n, lq, m_, l_ * p_))
    assert torch.equal(out_core, out_mod)
    gout = torch.ones_like(out_mod)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, sh, _lsi(sh), loc, aw, gout, 2)
    assert gv.shape == v.shape and gl.shape == loc.shape and ga.shape == aw.shape


# ---------------------------------------------------------------------------------------------- backward
@pytest.mark.parametrize("tag", ["kat", "oob", "c30", "c71"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_backward_matches_reference_autograd_goldens(tag, dtype):
    """lwdetr_msda_backward vs autograd through the reference's PyTorch core (tests/golden/msda_op_grad_kat.npz) and vs the
    C restatement of the reference's col2im kernel on the same inputs."""
    from helpers import load_golden
    from lwdetr_amd.ops.functions import ms_deform_attn_backward
    from oracle import msda_c
    g = load_golden("msda_op_grad_kat")
    shapes = torch.from_numpy(g[f"{tag}_shapes"]).to(DEV)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1])).contiguous()
    value, loc, aw, go = (torch.from_numpy(g[f"{tag}_{k}"]).to(dtype).to(DEV).contiguous() for k in ("value", "loc", "aw", "grad_out"))
    gv, gl, ga = ms_deform_attn_backward(value, shapes, lsi, loc, aw, go, 64)
    tol = 1e-9 if dtype == torch.float64 else 2e-4
    for ours, key in ((gv, "grad_value"), (gl, "grad_loc"), (ga, "grad_aw")):
        ref = g[f"{tag}_{key}"]
        assert np.abs(ours.double().cpu().numpy() - ref).max() <= tol * max(1.0, float(np.abs(ref).max())), key
    cv, cl, ca = msda_c.msda_backward(*(t.cpu().numpy() for t in (value,)), g[f"{tag}_shapes"],
                                      *(t.cpu().numpy() for t in (loc, aw, go)))
    ctol = 1e-12 if dtype == torch.float64 else 2e-5
    assert np.abs(gv.cpu().numpy() - cv).max() <= ctol * max(1.0, float(np.abs(cv).max()))
    assert np.abs(gl.cpu().numpy() - cl).max() <= ctol * 50 * max(1.0, float(np.abs(cl).max()))
    assert np.abs(ga.cpu().numpy() - ca).max() <= ctol * 50 * max(1.0, float(np.abs(ca).max()))


@pytest.mark.parametrize("channels", [30, 32, 64, 71, 1025])
def test_function_gradcheck_double(channels):
    """The reference's own gradient test (models/ops/test.py:89-112): numerical vs analytical gradients of
    MSDeformAttnFunction in double precision, same shapes and channel counts."""
    from lwdetr_amd.ops.functions import MSDeformAttnFunction
    n, m, lq, l, p = 1, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long, device=DEV)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    s = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    value = (torch.rand(n, s, m, channels, device=DEV) * 0.01).double().requires_grad_(True)
    loc = torch.rand(n, lq, m, l, p, 2, device=DEV).double().requires_grad_(True)
    aw = torch.rand(n, lq, m, l, p, device=DEV) + 1e-5
    aw = (aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)).double().requires_grad_(True)
    assert torch.autograd.gradcheck(MSDeformAttnFunction.apply, (value, shapes, lsi, loc, aw, 2), nondet_tol=1e-12)


@pytest.mark.parametrize("channels", [2048, 3096])
def test_function_gradients_large_channel_counts(channels):
    """The remaining channel counts of the reference's gradient test (models/ops/test.py:111-112: 1025, 2048, 3096; 1025 goes
    through gradcheck above). A numerical Jacobian over 2 x 30 x 2 x 3096 value elements is minutes of Python loop, so here the
    analytical gradients of the HIP backward are compared, in double precision, with autograd through the reference's PyTorch
    core (the oracle restatement) on the same shapes and seed as the reference's test."""
    from lwdetr_amd.ops.functions import MSDeformAttnFunction
    from oracle import lwdetr_torch as O
    n, m, lq, l, p = 1, 2, 2, 2, 2
    hw = [(6, 4), (3, 2)]
    shapes = torch.as_tensor(hw, dtype=torch.long, device=DEV)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    s = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    value = (torch.rand(n, s, m, channels, device=DEV) * 0.01).double().requires_grad_(True)
    loc = torch.rand(n, lq, m, l, p, 2, device=DEV).double().requires_grad_(True)
    aw = torch.rand(n, lq, m, l, p, device=DEV) + 1e-5
    aw = (aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)).double().requires_grad_(True)
    out = MSDeformAttnFunction.apply(value, shapes, lsi, loc, aw, 2)
    go = torch.rand(out.shape, device=DEV, dtype=torch.float64)
    gv, gl, ga = torch.autograd.grad(out, (value, loc, aw), go)
    v2, l2, a2 = (t.detach().cpu().requires_grad_(True) for t in (value, loc, aw))
    ref = O.msda_core(v2, hw, l2, a2)
    assert (out.detach().cpu() - ref.detach()).abs().max().item() < 1e-12
    rv, rl, ra = torch.autograd.grad(ref, (v2, l2, a2), go.cpu())
    for g, r in ((gv, rv), (gl, rl), (ga, ra)):
        assert (g.cpu() - r).abs().max().item() <= 1e-10 * max(1.0, r.abs().max().item())


def test_backward_model_shapes_and_module_autograd():
    """Training-style use: gradients flow through the reference-compatible module into its parameters and inputs."""
    from lwdetr_amd.ops.functions import MSDeformAttnFunction
    b, q, m, d, p = 2, 300, 16, 16, 2
    shapes = torch.as_tensor([(40, 40)], dtype=torch.long, device=DEV)
    lsi = torch.zeros(1, dtype=torch.long, device=DEV)
    g = torch.Generator(device="cpu").manual_seed(5)
    value = torch.randn(b, 1600, m, d, generator=g).to(DEV).requires_grad_(True)
    loc = (torch.rand(b, q, m, 1, p, 2, generator=g) * 1.2 - 0.1).to(DEV).requires_grad_(True)
    aw = torch.rand(b, q, m, 1, p, generator=g).softmax(-1).to(DEV).requires_grad_(True)
    out = MSDeformAttnFunction.apply(value, shapes, lsi, loc, aw, 64)
    go = torch.randn(out.shape, generator=g).to(DEV)
    out.backward(go)
    from oracle import lwdetr_torch as O
    v2, l2, a2 = (t.detach().cpu().double().requires_grad_(True) for t in (value, loc, aw))
    ref = O.msda_core(v2, [(40, 40)], l2, a2)
    rv, rl, ra = torch.autograd.grad(ref, (v2, l2, a2), go.cpu().double())
    assert (value.grad.cpu().double() - rv).abs().max().item() < 1e-4
    assert (aw.grad.cpu().double() - ra).abs().max().item() < 1e-3
    assert (loc.grad.cpu().double() - rl).abs().max().item() < 1e-3 * max(1.0, rl.abs().max().item())
