"""GPU: the DEFAULT launch plan per configuration, pinned (round 6). The plan is chosen by ~20 shape rules in Python and C plus the LWDETR_*
switches; a rule that drifts (a threshold edited, a switch that changes its default, a kernel that silently stops being taken) changes what the
benchmark measures without failing any numerics test. This test runs one forward per configuration with the library's per-kernel profiling hooks
on and asserts (a) the op classes of the plan and their counts and (b) the launches per kernel class of one step - against the table below.
Regenerate the table with `python tests/test_gpu_default_plan.py` on a GPU box after a DELIBERATE plan change and say why in the commit."""
import collections
import json
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# (size, batch of ONE launch chain, resolution, dtype)
CONFIGS = [("tiny", 16, 640, "float16"), ("small", 16, 640, "float16"), ("medium", 32, 640, "bfloat16"), ("large", 16, 640, "float16"),
           ("xlarge", 16, 960, "float16"), ("small", 1, 640, "float16"), ("large", 1, 640, "float16")]

EXPECTED = json.load(open(os.path.join(ROOT, "tests", "golden", "default_plan.json"))) if os.path.exists(
    os.path.join(ROOT, "tests", "golden", "default_plan.json")) else {}


def plan_signature(size, batch, res, dtype):
    import lwdetr_amd
    from lwdetr_amd import _native
    from lwdetr_amd.synth import synth_images, synth_state_dict
    T = getattr(torch, dtype)
    dev = torch.device("cuda:0")
    model, _, post = lwdetr_amd.build_model(lwdetr_amd.get_args(size))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
    model = model.to(dev).to(T).eval()
    x = synth_images(batch, res, res, seed=5).to(dev).to(T)
    plan = model._plan(batch, res, res)
    ops = collections.Counter()
    for group in (plan.ops_backbone, plan.ops_enc, plan.ops_sel, plan.ops_dec):
        for op in group:
            ops[type(op).__name__] += 1
    plan.run(x)                                   # warm-up (packs, attribute calls)
    torch.cuda.synchronize()
    _native.prof_collect()
    _native.prof_enable(True)
    try:
        plan.run(x)
        torch.cuda.synchronize()
        prof = _native.prof_collect()
    finally:
        _native.prof_enable(False)
    n_pt = _native.lib().lwdetr_gemm_pt_count()
    plan.run(x)
    torch.cuda.synchronize()
    return {"ops": dict(sorted(ops.items())), "launches": {k: int(v["count"]) for k, v in sorted(prof.items())},
            "persistent_gemm_launches": int(_native.lib().lwdetr_gemm_pt_count() - n_pt)}


def key(cfg):
    return "{}_b{}_{}_{}".format(*cfg)


@pytest.mark.parametrize("cfg", CONFIGS, ids=key)
def test_default_plan_is_the_pinned_one(cfg):
    assert key(cfg) in EXPECTED, "tests/golden/default_plan.json has no entry for this configuration: regenerate it (see the module docstring)"
    got = plan_signature(*cfg)
    assert got == EXPECTED[key(cfg)], json.dumps({"got": got, "expected": EXPECTED[key(cfg)]}, indent=1)


if __name__ == "__main__":
    table = {key(c): plan_signature(*c) for c in CONFIGS}
    out = os.path.join(ROOT, "gpurun_out", "default_plan.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(table, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(table, indent=1, sort_keys=True))
