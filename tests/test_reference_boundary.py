"""Drop-in boundaries against the UNMODIFIED reference (build container only; skipped where /root/reference is absent,
i.e. on the GPU box). Each check runs tests/ref_boundary_driver.py in a child process so the reference's top-level
package names (models, util, datasets, engine) never leak into this test session."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="/root/reference is not present")


def _drive(what):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_boundary_driver.py"), what], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_b4_reference_op_file_binds_to_our_native_module():
    """models/ops/functions/ms_deform_attn_func.py:23 `import MultiScaleDeformableAttention as MSDA` (pybind names
    models/ops/src/vision.cpp:13-16) resolves to lw-detr_amd/compat, and the reference's own MSDeformAttnFunction calls
    straight into our operator (on host tensors it trips OUR argument contract, not an ImportError / AttributeError)."""
    d = _drive("b4")
    assert d["func_file"].startswith("/root/reference/")
    assert d["msda_file"].endswith(os.path.join("lw-detr_amd", "compat", "MultiScaleDeformableAttention.py"))
    assert d["forward_is_ours"] and d["backward_is_ours"]
    assert d["host_call"] == "RuntimeError: value must be a CUDA tensor"


def test_b2_reference_evaluate_drives_our_model_and_postprocess():
    """/root/reference/engine.py:93-164 run as-is over a fake loader with our build_model() objects (the CPU oracle stands
    in for the HIP forward), a stub criterion and a recording CocoEvaluator: NestedTensor in, reference dict out,
    PostProcess results keyed by image id; and f4 - lwdetr_amd.dist.to_coco_results / to_evaluator_update produce exactly
    what datasets/coco_eval.py:91-113,176-178 builds from the same detections."""
    d = _drive("evaluate")
    assert d["forward_types"] == ["NestedTensor", "NestedTensor"] and d["iou_types"] == ["bbox"] and d["n_updates"] == 2
    assert d["records_equal"] and d["n_records"] == 200 and d["update_equal"]
    assert "coco_eval_bbox" in d["stats_keys"]
    assert set(d["first_record"]) == {"image_id", "category_id", "bbox", "score"} and len(d["first_record"]["bbox"]) == 4
