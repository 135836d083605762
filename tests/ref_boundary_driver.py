"""TEST INFRASTRUCTURE. Runs in a child process of tests/test_reference_boundary.py (build container only: /root/reference
must exist): imports the UNMODIFIED reference next to this package and checks the drop-in boundaries SURVEY section 8(b)
names - B4 (the native module name), B2 (engine.evaluate drives our objects), f4 (COCO result records).
Prints one JSON object; the parent asserts on it."""
import importlib
import json
import os
import sys
import types
from argparse import Namespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def _shim_base():
    """torchvision / timm / fairscale stand-ins (oracle/ref_shims.py) but NOT MultiScaleDeformableAttention: that name must
    resolve to lw-detr_amd/compat exactly as INTEGRATION.md section 2 describes."""
    from oracle import ref_shims
    ref_shims._install_shims()
    sys.modules.pop("MultiScaleDeformableAttention", None)
    sys.path.insert(0, os.path.join(ROOT, "lw-detr_amd", "compat"))
    sys.path.insert(1, REF)


def check_b4():
    import torch
    _shim_base()
    f = importlib.import_module("models.ops.functions.ms_deform_attn_func")        # the reference's file, unmodified
    import lwdetr_amd
    from lwdetr_amd.ops import functions as ours
    out = {"func_file": f.__file__, "msda_file": f.MSDA.__file__,
           "forward_is_ours": f.MSDA.ms_deform_attn_forward is ours.ms_deform_attn_forward,
           "backward_is_ours": f.MSDA.ms_deform_attn_backward is ours.ms_deform_attn_backward}
    # the reference's own autograd Function reaches our entry point: host tensors hit OUR argument contract
    v = torch.rand(1, 6, 2, 8)
    shapes = torch.tensor([[2, 3]], dtype=torch.long)
    lsi = torch.tensor([0], dtype=torch.long)
    loc = torch.rand(1, 4, 2, 1, 2, 2)
    aw = torch.rand(1, 4, 2, 1, 2)
    try:
        f.MSDeformAttnFunction.apply(v, shapes, lsi, loc, aw, 64)
        out["host_call"] = "no error"
    except Exception as e:      # noqa: BLE001
        out["host_call"] = f"{type(e).__name__}: {e}"
    return out


def check_evaluate():
    import torch
    _shim_base()
    # pycocotools is not installed: datasets/coco_eval.py only needs the names at import time (the evaluator is stubbed)
    for name in ("pycocotools", "pycocotools.cocoeval", "pycocotools.coco", "pycocotools.mask"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["pycocotools.cocoeval"].COCOeval = object
    sys.modules["pycocotools.coco"].COCO = object
    sys.modules["pycocotools"].mask = sys.modules["pycocotools.mask"]
    pkg = types.ModuleType("datasets")                 # package shell: skips datasets/__init__.py (torchvision datasets)
    pkg.__path__ = [os.path.join(REF, "datasets")]
    sys.modules["datasets"] = pkg
    engine = importlib.import_module("engine")         # /root/reference/engine.py
    coco_eval = importlib.import_module("datasets.coco_eval")
    from util.misc import NestedTensor as RefNested

    import lwdetr_amd
    from lwdetr_amd import dist as D
    from lwdetr_amd.synth import synth_images, synth_state_dict
    from oracle import lwdetr_torch as O

    cfg = lwdetr_amd.get_args("tiny")
    model, _, post = lwdetr_amd.build_model(cfg)
    sd = synth_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd)
    calls = {"forward_types": [], "updates": []}

    def host_forward(samples):      # the CPU oracle stands in for the HIP forward: this test is about the call boundary
        calls["forward_types"].append(type(samples).__name__)
        return O.forward(sd, cfg, samples.tensors, samples.mask)
    model.forward = host_forward

    class Criterion:
        weight_dict = {"loss_ce": 2.0}

        def eval(self):
            return self

        def __call__(self, outputs, targets):
            assert set(outputs) >= {"pred_logits", "pred_boxes", "aux_outputs", "enc_outputs"}
            return {"loss_ce": outputs["pred_logits"].mean() * 0 + 1.0, "class_error": torch.tensor(0.0)}

    class Evaluator:
        def __init__(self, base_ds, iou_types):
            calls["iou_types"] = list(iou_types)
            self.coco_eval = {"bbox": Namespace(stats=torch.zeros(12))}

        def update(self, res):
            calls["updates"].append(res)

        def synchronize_between_processes(self):
            pass

        def accumulate(self):
            pass

        def summarize(self):
            pass
    engine.CocoEvaluator = Evaluator

    imgs = synth_images(2, 192, 256, seed=7)
    loader = []
    for i in range(2):
        x = imgs[i:i + 1]
        samples = RefNested(x, torch.zeros(1, 192, 256, dtype=torch.bool))
        loader.append((samples, [{"image_id": torch.tensor(100 + i), "orig_size": torch.tensor([480, 640])}]))
    stats, _ = engine.evaluate(model, Criterion(), post, loader, None, torch.device("cpu"), Namespace(fp16_eval=False))
    res = {}
    for u in calls["updates"]:
        res.update(u)
    # f4: the reference's own record builder vs ours, on exactly the dict its loop produced
    ref_records = coco_eval.CocoEvaluator.prepare_for_coco_detection(None, res)
    ids = torch.tensor(sorted(res))
    scores = torch.stack([res[int(i)]["scores"] for i in ids])
    labels = torch.stack([res[int(i)]["labels"] for i in ids])
    boxes = torch.stack([res[int(i)]["boxes"] for i in ids])
    ours = D.to_coco_results(ids, scores, labels, boxes)
    upd = D.to_evaluator_update(ids, scores, labels, boxes)
    same_update = sorted(upd) == sorted(res) and all(
        torch.equal(upd[k][f], res[k][f]) for k in res for f in ("scores", "labels", "boxes"))
    return {"forward_types": calls["forward_types"], "iou_types": calls["iou_types"], "n_updates": len(calls["updates"]),
            "records_equal": ref_records == ours, "n_records": len(ours), "update_equal": bool(same_update),
            "stats_keys": sorted(stats), "first_record": ours[0]}


if __name__ == "__main__":
    print(json.dumps(check_b4() if sys.argv[1] == "b4" else check_evaluate()))
