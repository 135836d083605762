"""CPU: the torch restatement oracle (oracle/lwdetr_torch.py) is pinned against reference-generated goldens."""
import numpy as np
import pytest
import torch

import lwdetr_amd
from oracle import lwdetr_torch as O
from helpers import CASES, case_batch, golden_state_dict, load_golden, sample_idx

FAST = ["tiny_640", "tiny_192x256", "small_padded", "large_padded", "medium_640"]
SLOW = ["small_640", "large_640", "xlarge_640", "xlarge_960"]


def _check(name):
    g = load_golden(name)
    size, images, mask = case_batch(name)
    cfg = lwdetr_amd.get_args(size)
    sd = golden_state_dict(g)
    col = {}
    with torch.no_grad():
        out = O.forward(sd, cfg, images, mask, collect=col)
    # two-stage selection identical (order included) - fp32 noise must not flip top-k on these inputs
    assert np.array_equal(out["topk_idx"].numpy(), g["topk_idx"])
    tol = dict(rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["pred_logits"].numpy(), g["pred_logits"], **tol)
    np.testing.assert_allclose(out["pred_boxes"].numpy(), g["pred_boxes"], **tol)
    np.testing.assert_allclose(out["enc_outputs"]["pred_logits"].numpy(), g["enc_logits"], **tol)
    np.testing.assert_allclose(out["enc_outputs"]["pred_boxes"].numpy(), g["enc_boxes"], **tol)
    for j, aux in enumerate(out["aux_outputs"]):
        np.testing.assert_allclose(aux["pred_logits"].numpy(), g[f"aux{j}_logits"], **tol)
        np.testing.assert_allclose(aux["pred_boxes"].numpy(), g[f"aux{j}_boxes"], **tol)
    for li in range(len(cfg.projector_scale)):
        flat = col[f"proj.level{li}"].reshape(-1).numpy()
        np.testing.assert_allclose(flat[sample_idx(flat.size)], g[f"stage.proj.level{li}"], **tol)
    res = O.postprocess(out, torch.tensor([[480.0, 640.0]] * images.shape[0]), cfg.num_select)
    np.testing.assert_allclose(torch.stack([r["scores"] for r in res]).numpy(), g["post_scores"], atol=2e-6)
    assert np.array_equal(torch.stack([r["labels"] for r in res]).numpy(), g["post_labels"])
    np.testing.assert_allclose(torch.stack([r["boxes"] for r in res]).numpy(), g["post_boxes"], atol=2e-3)


@pytest.mark.parametrize("name", FAST)
def test_oracle_matches_reference_golden(name):
    _check(name)


@pytest.mark.parametrize("name", SLOW)
def test_oracle_matches_reference_golden_large(name):
    _check(name)


def test_golden_cases_table_consistent():
    for name, (size, dims) in CASES.items():
        g = load_golden(name)
        assert [tuple(x) for x in g["image_hw"].tolist()] == dims
        assert g["pred_logits"].shape[0] == len(dims)
