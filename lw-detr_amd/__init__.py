"""lwdetr_amd - MI355X-native LW-DETR inference forward path.

Public surface mirrors the reference's model API (``models/__init__.py:16-17``):
``build_model(args) -> (model, criterion, postprocessors)``; the operator API lives in
``lwdetr_amd.ops`` (``MSDeformAttnFunction``, ``MSDeformAttn``, ``ms_deform_attn_forward``).
"""
from .configs import SIZES, get_args  # noqa: F401

__version__ = "0.1.0"


def build_model(args):
    from .models import build_model as _build
    return _build(args)
