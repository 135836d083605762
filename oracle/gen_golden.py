"""TEST INFRASTRUCTURE ONLY - generate ``tests/golden/*.npz`` by running the UNMODIFIED reference.

Run in the build container (``/root/reference`` is mounted there, not on the GPU box):

    python -m oracle.gen_golden            # all cases
    python -m oracle.gen_golden tiny_640   # one case

For every case the reference model (``models.build_model`` via ``oracle/ref_shims.py``) is loaded with the
deterministic synthetic state dict of ``lwdetr_amd.synth`` (``strict=True`` - this also pins the state-dict key
set) and run in fp32 on CPU with the reference's own PyTorch deformable-attention core. Stored per case:
final / aux / encoder outputs in full, the two-stage top-k indices, PostProcess results, and strided samples
of the intermediate stages (ViT taps, projector levels, decoder layers) captured with forward hooks.
The op-level known-answer vectors restate ``models/ops/test.py:27-60`` (seed 3) on CPU.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import lwdetr_amd  # noqa: E402
from lwdetr_amd.synth import synth_images, synth_state_dict  # noqa: E402
from oracle.ref_shims import build_reference_model, import_reference  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# name -> (size, list of (h, w) per image; images smaller than the max are zero padded and masked)
CASES = {
    "tiny_640": ("tiny", [(640, 640)]),
    "small_640": ("small", [(640, 640), (640, 640)]),
    "medium_640": ("medium", [(640, 640)]),
    "large_640": ("large", [(640, 640)]),
    "xlarge_640": ("xlarge", [(640, 640)]),
    "xlarge_960": ("xlarge", [(960, 960)]),
    "tiny_192x256": ("tiny", [(192, 256), (192, 256)]),
    "small_padded": ("small", [(448, 512), (320, 384)]),
    "large_padded": ("large", [(384, 320), (256, 320)]),
}
MAX_SAMPLES = 4096


def sample_idx(n: int) -> np.ndarray:
    """Deterministic strided sample positions of a flattened tensor (shared with the tests)."""
    if n <= MAX_SAMPLES:
        return np.arange(n)
    stride = -(-n // MAX_SAMPLES)
    stride += 1 - (stride % 2)          # odd stride: walks across rows and channels
    return np.arange(0, n, stride)[:MAX_SAMPLES]


def case_inputs(name):
    """Images (list of (3,h,w)) for a case; deterministic."""
    size, dims = CASES[name]
    hmax, wmax = max(d[0] for d in dims), max(d[1] for d in dims)
    full = synth_images(len(dims), hmax, wmax, seed=1234)
    return size, [full[i, :, :h, :w].clone() for i, (h, w) in enumerate(dims)]


def run_case(name: str):
    size, imgs = case_inputs(name)
    args = lwdetr_amd.get_args(size)
    model, post = build_reference_model(args)
    sd = synth_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd, strict=True)
    keys = np.array(sorted(sd.keys()))
    shapes = np.array([",".join(map(str, sd[k].shape)) for k in keys])

    store = {}

    def keep(tag, t):
        flat = t.detach().float().reshape(-1).numpy()
        store[f"stage.{tag}"] = flat[sample_idx(flat.size)]
        store[f"stageshape.{tag}"] = np.array(t.shape)

    enc = model.backbone[0].encoder
    def hook_list(prefix):
        def fn(_m, _i, o):
            for j, t in enumerate(o):
                keep(f"{prefix}{j}", t)
        return fn

    def hook_one(tag):
        def fn(_m, _i, o):
            keep(tag, o)
        return fn

    enc.register_forward_hook(hook_list("vit.tap"))
    model.backbone[0].projector.register_forward_hook(hook_list("proj.level"))
    for j, lay in enumerate(model.transformer.decoder.layers):
        lay.register_forward_hook(hook_one(f"dec.layer{j}.out"))
    topk_holder = {}
    orig_topk = torch.topk

    def spy_topk(inp, k, dim=-1, **kw):
        r = orig_topk(inp, k, dim=dim, **kw)
        if "enc" not in topk_holder and inp.dim() == 2 and k == args.num_queries:
            topk_holder["enc"] = (inp.detach().clone(), r[1].detach().clone())
        return r

    torch.topk = spy_topk
    t0 = time.time()
    try:
        with torch.no_grad():
            out = model(imgs if len({tuple(i.shape) for i in imgs}) > 1 else torch.stack(imgs))
    finally:
        torch.topk = orig_topk
    dt = time.time() - t0

    hmax, wmax = max(i.shape[1] for i in imgs), max(i.shape[2] for i in imgs)
    sizes = torch.tensor([[480.0, 640.0]] * len(imgs))
    with torch.no_grad():
        res = post["bbox"](out, sizes)
    store.update({
        "pred_logits": out["pred_logits"].numpy(), "pred_boxes": out["pred_boxes"].numpy(),
        "enc_logits": out["enc_outputs"]["pred_logits"].numpy(),
        "enc_boxes": out["enc_outputs"]["pred_boxes"].numpy(),
        "topk_idx": topk_holder["enc"][1].numpy(), "enc_class_max": topk_holder["enc"][0].numpy(),
        "post_scores": torch.stack([r["scores"] for r in res]).numpy(),
        "post_labels": torch.stack([r["labels"] for r in res]).numpy(),
        "post_boxes": torch.stack([r["boxes"] for r in res]).numpy(),
        "image_hw": np.array([[i.shape[1], i.shape[2]] for i in imgs]), "padded_hw": np.array([hmax, wmax]),
        "sd_keys": keys, "sd_shapes": shapes,
    })
    for j, aux in enumerate(out["aux_outputs"]):
        store[f"aux{j}_logits"] = aux["pred_logits"].numpy()
        store[f"aux{j}_boxes"] = aux["pred_boxes"].numpy()
    path = os.path.join(GOLDEN_DIR, f"{name}.npz")
    np.savez_compressed(path, **store)
    print(f"{name}: reference forward {dt:.2f}s -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def run_op_kat():
    """Known-answer vectors of models/ops/test.py:27-60 (shapes :27-31, seed :34, generators :39-42), on CPU."""
    import_reference()
    from models.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
    n, m, d, lq, l, p = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    s = int((shapes[:, 0] * shapes[:, 1]).sum())
    torch.manual_seed(3)
    store = {"shapes": shapes.numpy()}
    for tag in ("double", "float"):            # same draw order as the reference test script
        value = torch.rand(n, s, m, d) * 0.01
        loc = torch.rand(n, lq, m, l, p, 2)
        aw = torch.rand(n, lq, m, l, p) + 1e-5
        aw /= aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
        if tag == "double":
            value, loc, aw = value.double(), loc.double(), aw.double()
        out = ms_deform_attn_core_pytorch(value.permute(0, 2, 3, 1), shapes, loc, aw)
        store.update({f"{tag}_value": value.numpy(), f"{tag}_loc": loc.numpy(), f"{tag}_aw": aw.numpy(),
                      f"{tag}_out": out.numpy()})
    # a larger case with out-of-range locations (zero-padding branch, .cuh:288 and :56-78), model-like shapes
    g = torch.Generator().manual_seed(7)
    shapes2 = torch.as_tensor([(20, 12), (10, 6)], dtype=torch.long)
    s2 = int((shapes2[:, 0] * shapes2[:, 1]).sum())
    value = torch.randn(2, s2, 4, 16, generator=g)
    loc = torch.rand(2, 37, 4, 2, 4, 2, generator=g) * 1.4 - 0.2
    aw = torch.rand(2, 37, 4, 2, 4, generator=g).flatten(-2).softmax(-1).view(2, 37, 4, 2, 4)
    out = ms_deform_attn_core_pytorch(value.permute(0, 2, 3, 1).contiguous(), shapes2, loc, aw)
    store.update({"oob_shapes": shapes2.numpy(), "oob_value": value.numpy(), "oob_loc": loc.numpy(),
                  "oob_aw": aw.numpy(), "oob_out": out.numpy()})
    path = os.path.join(GOLDEN_DIR, "msda_op_kat.npz")
    np.savez_compressed(path, **store)
    print("op KAT ->", path, "double out:", np.round(store["double_out"], 4).tolist())


def run_op_grad_kat():
    """Gradients of the reference's own differentiable CPU arithmetic for the op (``ms_deform_attn_core_pytorch`` through
    autograd, double precision): the values ``ms_deform_attn_backward`` (ms_deform_attn_cuda.cu:83-153) must reproduce.
    Shapes of models/ops/test.py:27-31 plus a model-like case with out-of-range locations and channel counts of the
    reference's gradient checks (test.py:111: 30, 32, 64, 71)."""
    import_reference()
    from models.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
    store = {}
    g = torch.Generator().manual_seed(11)
    cases = {"kat": (1, 2, 2, 2, [(6, 4), (3, 2)], 2), "oob": (2, 4, 16, 37, [(20, 12), (10, 6)], 4),
             "c30": (1, 2, 30, 5, [(6, 4), (3, 2)], 2), "c71": (1, 2, 71, 3, [(6, 4), (3, 2)], 2)}
    for tag, (n, m, d, lq, sh, p) in cases.items():
        shapes = torch.as_tensor(sh, dtype=torch.long)
        l, s = len(sh), int((shapes[:, 0] * shapes[:, 1]).sum())
        value = (torch.rand(n, s, m, d, generator=g, dtype=torch.float64) * 0.01 if tag == "kat"
                 else torch.randn(n, s, m, d, generator=g, dtype=torch.float64)).requires_grad_(True)
        loc = torch.rand(n, lq, m, l, p, 2, generator=g, dtype=torch.float64)
        if tag == "oob":
            loc = loc * 1.4 - 0.2
        loc.requires_grad_(True)
        aw = torch.rand(n, lq, m, l, p, generator=g, dtype=torch.float64) + 1e-5
        aw = (aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)).requires_grad_(True)
        out = ms_deform_attn_core_pytorch(value.permute(0, 2, 3, 1).contiguous(), shapes, loc, aw)
        go = torch.randn(out.shape, generator=g, dtype=torch.float64)
        gv, gl, ga = torch.autograd.grad(out, (value, loc, aw), go)
        for k, t in (("shapes", shapes), ("value", value), ("loc", loc), ("aw", aw), ("grad_out", go), ("grad_value", gv),
                     ("grad_loc", gl), ("grad_aw", ga)):
            store[f"{tag}_{k}"] = t.detach().numpy()
    path = os.path.join(GOLDEN_DIR, "msda_op_grad_kat.npz")
    np.savez_compressed(path, **store)
    print("op gradient KAT ->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    which = sys.argv[1:] or (["op_kat"] + list(CASES))
    for w in which:
        if w == "op_kat":
            run_op_kat()
        elif w == "op_grad_kat":
            run_op_grad_kat()
        else:
            run_case(w)
