"""GPU: the sorted radix top-k kernel, the class-max kernel, the fused PostProcess kernel and the HIP-graph replay.

Expected order of the top-k = descending value, ties by ascending index = a STABLE descending sort (computed on the host);
torch.topk itself leaves the order of ties unspecified, so it is only used where values are distinct."""
import pytest
import torch

import lwdetr_amd
import lwdetr_amd.models
from lwdetr_amd import _native
from lwdetr_amd.synth import synth_images, synth_state_dict

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _topk(x, k):
    x = x.contiguous()
    b, n = x.shape
    idx = torch.empty(b, k, dtype=torch.int64, device=x.device)
    val = torch.empty(b, k, dtype=torch.float32, device=x.device)
    rc = _native.lib().lwdetr_topk(x.data_ptr(), b, n, k, idx.data_ptr(), val.data_ptr(), _native.dtype_code(x.dtype),
                                   _native.stream_ptr(x.device))
    _native.check(rc, "lwdetr_topk")
    return val, idx


def _expected(x, k):
    v, i = torch.sort(x.float().cpu(), dim=1, descending=True, stable=True)
    return v[:, :k], i[:, :k]


@pytest.mark.parametrize("shape_k", [((3, 8400), 300), ((2, 27300), 300), ((1, 18900), 300), ((4, 300), 300), ((2, 1), 1),
                                     ((1, 5000), 1024), ((33, 2100), 100)])
@pytest.mark.parametrize("kind", ["distinct", "quantised", "constant", "special"])
def test_topk_sorted_with_ties(shape_k, kind):
    (b, n), k = shape_k
    g = torch.Generator().manual_seed(n + k)
    x = torch.randn(b, n, generator=g) * 3.0 - 2.0
    if kind == "quantised":
        x = (x * 4).round() / 4 + 0.0             # heavy ties, including at the k-th value (+0.0: no negative zeros)
    elif kind == "constant":
        x = torch.full((b, n), -1.25)             # every element ties: the index order decides alone
    elif kind == "special":
        x[:, ::7] = 0.0
        x[:, 1::11] = -0.0
        x[:, 2::13] = float("inf")
        x[:, 3::17] = float("-inf")
    val, idx = _topk(x.to(DEV), k)
    ev, ei = _expected(x, k)
    assert torch.equal(val.cpu(), ev)
    if kind == "special":                         # +0 / -0 compare equal for torch's sort but are ordered by the kernel
        assert torch.equal(torch.gather(x, 1, idx.cpu()), ev)
        assert all(len(set(r.tolist())) == k for r in idx.cpu())
    else:
        assert torch.equal(idx.cpu(), ei)
    if kind == "distinct":
        assert torch.equal(idx.cpu(), torch.topk(x, k, dim=1)[1])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_topk_16bit_inputs(dtype):
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(4, 27300, generator=g) * 2 - 3).to(dtype)
    val, idx = _topk(x.to(DEV), 300)
    ev, ei = _expected(x, 300)
    assert torch.equal(val.cpu(), ev) and torch.equal(idx.cpu(), ei)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows_cols_ld", [(8400, 91, 92), (1000, 91, 96), (37, 20, 20), (5, 366, 368), (16, 3, 4)])
def test_rowmax(dtype, rows_cols_ld):
    rows, cols, ld = rows_cols_ld
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, ld, generator=g).to(dtype).to(DEV)
    x[:, cols:] = 100.0                           # padding columns must not be read into the maximum
    out = torch.empty(rows, dtype=torch.float32, device=DEV)
    rc = _native.lib().lwdetr_rowmax(x.data_ptr(), ld, rows, cols, out.data_ptr(), _native.dtype_code(dtype), _native.stream_ptr(DEV))
    _native.check(rc, "lwdetr_rowmax")
    assert torch.equal(out, x[:, :cols].float().max(1)[0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("b_nq_c", [(2, 300, 91), (5, 100, 20), (1, 300, 366)])
def test_postprocess_kernel_matches_tensor_ops(dtype, b_nq_c):
    b, nq, c = b_nq_c
    g = torch.Generator().manual_seed(c)
    logits = (torch.randn(b, nq, c, generator=g) * 2 - 3).to(dtype)
    boxes = torch.rand(b, nq, 4, generator=g).to(dtype)
    boxes[:, ::9, 2] = -0.05                      # negative widths are clamped (util/box_ops.py:21-25)
    sizes = torch.tensor([[480.0, 640.0], [333.0, 500.0], [800.0, 1216.0], [64.0, 64.0], [427.0, 640.0]])[:b]
    post = lwdetr_amd.models.PostProcess(min(300, nq * c))
    s, l, bx = post.select(logits.to(DEV), boxes.to(DEV), sizes.to(DEV))
    # expectation: the same tensor ops as the reference, with the tie order made explicit (stable sort of the logits)
    k = post.num_select
    lv, li = torch.sort(logits.float().view(b, -1), dim=1, descending=True, stable=True)
    lv, li = lv[:, :k], li[:, :k]
    es = lv.sigmoid().to(dtype).float()
    cx, cy, w, h = boxes.unbind(-1)
    w, h = w.clamp(min=0), h.clamp(min=0)
    xyxy = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)          # in `dtype`, as the reference
    exy = torch.gather(xyxy, 1, (li // c).unsqueeze(-1).repeat(1, 1, 4)).float()
    exy = exy * torch.stack([sizes[:, 1], sizes[:, 0], sizes[:, 1], sizes[:, 0]], 1)[:, None, :]
    assert s.dtype == torch.float32 and l.dtype == torch.int64 and bx.dtype == torch.float32
    ulp = {torch.float32: 5e-7, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]
    assert (s.cpu() - es).abs().max().item() <= ulp
    assert torch.equal(l.cpu(), li % c)
    assert (bx.cpu() - exy).abs().max().item() <= ulp * 1300
    # the packed form (the record of the detection all-gather) carries exactly the same values
    from lwdetr_amd.dist import pack_detections
    packed = post.select_packed(logits.to(DEV), boxes.to(DEV), sizes.to(DEV))
    assert packed.shape == (b, k, 6) and torch.equal(packed, pack_detections(s, l, bx))
    # the module form returns the reference's list of dicts, scores in the logits' dtype
    res = post({"pred_logits": logits.to(DEV), "pred_boxes": boxes.to(DEV)}, sizes.to(DEV))
    assert len(res) == b and res[0]["scores"].dtype == dtype and res[0]["boxes"].shape == (k, 4)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_hip_graph_replay_is_bit_identical_to_eager(dtype):
    cfg = lwdetr_amd.get_args("tiny")
    model, _, post = lwdetr_amd.build_model(cfg)
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
    model = model.to(DEV).to(dtype).eval()
    x0 = synth_images(1, 320, 320, seed=1).to(DEV).to(dtype)
    x1 = synth_images(1, 320, 320, seed=2).to(DEV).to(dtype)
    sizes = torch.tensor([[480.0, 640.0]], device=DEV)
    graphed = model.capture(x0, postprocess=post["bbox"], target_sizes=sizes)
    for x in (x1, x0, x1):
        out_g, det_g = graphed(x)
        out_g = {k: out_g[k].clone() for k in ("pred_logits", "pred_boxes")}
        det_g = [t.clone() for t in det_g]
        out_e = model(x)
        det_e = post["bbox"].select(out_e["pred_logits"], out_e["pred_boxes"], sizes)
        assert torch.equal(out_g["pred_logits"], out_e["pred_logits"]) and torch.equal(out_g["pred_boxes"], out_e["pred_boxes"])
        assert all(torch.equal(a, b) for a, b in zip(det_g, det_e))
