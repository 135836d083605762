"""Make the *unmodified* reference at /root/reference importable in this container (never on the GPU box).

TEST INFRASTRUCTURE ONLY. The reference imports four packages that are absent from the image; tiny stand-ins
are injected into ``sys.modules`` (recipe: SURVEY.md section 9.1):
  * ``torchvision``            - ``util/misc.py:31-33``, ``util/box_ops.py:18``
  * ``timm.models.layers``     - ``models/backbone/vit.py:21`` (DropPath, Mlp, trunc_normal_)
  * ``fairscale.nn.checkpoint``- ``models/backbone/vit.py:20`` (checkpoint_wrapper, disabled by backbone.py:71)
  * ``MultiScaleDeformableAttention`` - ``models/ops/functions/ms_deform_attn_func.py:23`` (CUDA-only ext)
Used only by ``oracle/gen_golden.py`` to produce the committed fixtures under ``tests/golden``.
"""
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models"))


def _install_shims():
    if "timm" in sys.modules and getattr(sys.modules["timm"], "_lwdetr_shim", False):
        return
    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.25.0"
    tv._is_tracing = lambda: False
    tv_ops = types.ModuleType("torchvision.ops")
    tv_boxes = types.ModuleType("torchvision.ops.boxes")
    tv_boxes.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    tv_misc = types.ModuleType("torchvision.ops.misc")
    tv.ops, tv_ops.boxes, tv_ops.misc = tv_ops, tv_boxes, tv_misc
    sys.modules.update({"torchvision": tv, "torchvision.ops": tv_ops,
                        "torchvision.ops.boxes": tv_boxes, "torchvision.ops.misc": tv_misc})

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            return x

    class Mlp(nn.Module):  # timm.models.layers.Mlp: fc1 -> act -> fc2 (drop = 0)
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
            super().__init__()
            self.fc1 = nn.Linear(in_features, hidden_features or in_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    timm = types.ModuleType("timm")
    timm._lwdetr_shim = True
    timm_models = types.ModuleType("timm.models")
    timm_layers = types.ModuleType("timm.models.layers")
    timm_layers.DropPath, timm_layers.Mlp, timm_layers.trunc_normal_ = DropPath, Mlp, nn.init.trunc_normal_
    timm.models, timm_models.layers = timm_models, timm_layers
    sys.modules.update({"timm": timm, "timm.models": timm_models, "timm.models.layers": timm_layers})

    fs = types.ModuleType("fairscale")
    fs_nn = types.ModuleType("fairscale.nn")
    fs_ck = types.ModuleType("fairscale.nn.checkpoint")
    fs_ck.checkpoint_wrapper = lambda m: m
    fs.nn, fs_nn.checkpoint = fs_nn, fs_ck
    sys.modules.update({"fairscale": fs, "fairscale.nn": fs_nn, "fairscale.nn.checkpoint": fs_ck})

    msda = types.ModuleType("MultiScaleDeformableAttention")

    def _no_native(*a, **k):
        raise RuntimeError("reference CUDA extension is not available; the oracle uses the reference's "
                           "own ms_deform_attn_core_pytorch path (_export=True)")
    msda.ms_deform_attn_forward = msda.ms_deform_attn_backward = _no_native
    sys.modules["MultiScaleDeformableAttention"] = msda


def import_reference():
    """Return the reference's ``models`` package (imported from /root/reference through the shims)."""
    if not reference_available():
        raise RuntimeError(f"{REFERENCE_ROOT} is not present (it only exists in the build container)")
    _install_shims()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import models  # noqa: the reference's package
    return models


def build_reference_model(args):
    """Unmodified reference LWDETR in eval mode, MSDeformAttn on its own PyTorch (grid_sample) core.

    Oracle mode (A) of SURVEY.md section 8c: ``_export=True`` only on ``MSDeformAttn`` modules
    (``models/ops/modules/ms_deform_attn.py:133-136``), so ``LWDETR.forward`` keeps its dict API, masks and
    valid_ratios and works at any resolution that is a multiple of 64.
    """
    models = import_reference()
    model, _criterion, post = models.build_model(args)
    model.eval()
    from models.ops.modules import MSDeformAttn
    for m in model.modules():
        if isinstance(m, MSDeformAttn):
            m._export = True
    return model, post
