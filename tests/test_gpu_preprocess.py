"""GPU: the input-side kernels (uint8 HWC -> Pillow-exact square resize -> ToTensor -> Normalize -> NCHW) against the oracle
(bit-exact: integer resize, table-driven normalisation) and against the committed Pillow goldens."""
import os

import numpy as np
import pytest
import torch

import lwdetr_amd
from helpers import ROOT
from lwdetr_amd.preprocess import SquareResizeNormalize, infer_transforms
from oracle import preprocess_ref as P

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_matches_pillow_goldens(dtype):
    g = np.load(os.path.join(ROOT, "tests", "golden", "preprocess.npz"))
    n = sum(1 for k in g.files if k.startswith("in_"))
    for i in range(n):
        s = int(g[f"size_{i}"])
        out, sizes = SquareResizeNormalize(s, dtype=dtype, device=DEV)([torch.from_numpy(g[f"in_{i}"]).to(DEV)])
        ref = P.to_tensor_normalize(g[f"out_{i}"], dtype)          # Pillow's pixels through the reference's f32 arithmetic
        assert out.shape == (1, 3, s, s) and out.dtype == dtype
        assert torch.equal(out[0].cpu(), ref), i
        assert sizes.tolist() == [[float(g[f"in_{i}"].shape[0]), float(g[f"in_{i}"].shape[1])]]


def test_mixed_size_batch_strided_rows_and_upscaling_bit_exact_vs_oracle():
    rng = np.random.default_rng(3)
    shapes = [(480, 640), (427, 640), (640, 480), (333, 500), (64, 48), (720, 1280), (640, 640)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    dev_imgs = [torch.from_numpy(im).to(DEV) for im in imgs]
    wide = torch.zeros(333, 600, 3, dtype=torch.uint8, device=DEV)      # image 3 as a view with padded rows
    wide[:, :500] = dev_imgs[3]
    dev_imgs[3] = wide[:, :500]
    wide0 = torch.zeros(480, 701, 3, dtype=torch.uint8, device=DEV)     # image 0 (width == S: read in place) with odd row pitch
    wide0[:, :640] = dev_imgs[0]
    dev_imgs[0] = wide0[:, :640]
    tf = infer_transforms(640, dtype=torch.float32, device=DEV)
    out, sizes = tf(dev_imgs)
    ref, ref_sizes = P.preprocess(imgs, 640, torch.float32)
    assert torch.equal(out.cpu(), ref) and torch.equal(sizes.cpu(), ref_sizes)
    out2, _ = tf(dev_imgs[::-1])                                        # cached tables, different order
    assert torch.equal(out2.cpu(), ref.flip(0))
    # a (B, H, W, 3) tensor and host tensors are accepted too
    stack = torch.from_numpy(np.stack([imgs[0], imgs[0][::-1].copy()]))
    out3, _ = tf(stack)
    assert torch.equal(out3[0].cpu(), ref[0])


def test_table_cache_is_per_axis_length_and_bounded():
    """resample tables are cached per axis length (x: width, y: height) in one device buffer that is appended to
    until it is full and then started over; a stream of distinct sizes neither re-uploads old tables nor grows without bound, a
    wrapped cache stays exact, and a single batch whose tables exceed the buffer makes it grow instead of failing."""
    rng = np.random.default_rng(5)
    tf = infer_transforms(320, dtype=torch.float32, device=DEV)
    tf._table_dev = torch.empty(12000, dtype=torch.int32, device=DEV)        # small: the stream below wraps several times
    first = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    ref0, _ = P.preprocess([first], 320, torch.float32)
    ptr, wraps, last_len = tf._table_dev.data_ptr(), 0, 0
    for k in range(12):
        h, w = 64 + 7 * k, 80 + 5 * k
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out, _ = tf([torch.from_numpy(im).to(DEV), torch.from_numpy(first).to(DEV)])
        ref, _ = P.preprocess([im], 320, torch.float32)
        assert torch.equal(out[0].cpu(), ref[0]) and torch.equal(out[1].cpu(), ref0[0]), k
        wraps += tf._table_len < last_len
        last_len = tf._table_len
        assert tf._table_len <= tf._table_dev.numel()
    assert wraps >= 1 and tf._table_dev.data_ptr() == ptr      # started over, same buffer: nothing re-allocated or re-concatenated
    # one batch with more distinct axis lengths than the buffer holds: the buffer grows, the batch is served, results exact
    many = [rng.integers(0, 256, (200 + 3 * i, 260 + 5 * i, 3), dtype=np.uint8) for i in range(10)]
    out, _ = tf([torch.from_numpy(m).to(DEV) for m in many])
    ref, _ = P.preprocess(many, 320, torch.float32)
    assert torch.equal(out.cpu(), ref) and tf._table_dev.numel() > 12000


def test_feeds_the_model_end_to_end():
    """uint8 frames -> preprocess -> LWDETR -> PostProcess, all on the device; fp32 result equals the oracle pipeline."""
    from lwdetr_amd.synth import synth_state_dict
    from oracle import lwdetr_torch as O
    cfg = lwdetr_amd.get_args("tiny")
    model, _, post = lwdetr_amd.build_model(cfg)
    sd = synth_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    rng = np.random.default_rng(11)
    frames = [rng.integers(0, 256, (240, 320, 3), dtype=np.uint8), rng.integers(0, 256, (200, 300, 3), dtype=np.uint8)]
    batch, sizes = SquareResizeNormalize(256, dtype=torch.float32, device=DEV)([torch.from_numpy(f) for f in frames])
    ref_batch, ref_sizes = P.preprocess(frames, 256, torch.float32)
    assert torch.equal(batch.cpu(), ref_batch)
    with torch.no_grad():
        exp = O.forward(sd, cfg, ref_batch)
    out = model(batch, _forced_topk=exp["topk_idx"])      # slot-wise comparison: same two-stage selection (near-ties)
    res = post["bbox"](out, sizes)
    assert (out["pred_boxes"].cpu() - exp["pred_boxes"]).abs().max().item() < 1e-3
    assert (out["pred_logits"].cpu() - exp["pred_logits"]).abs().max().item() < 1e-3
    assert len(res) == 2 and res[0]["boxes"].shape[-1] == 4
