"""Operator API of the reference (``models/ops/functions/__init__.py:13``, ``models/ops/modules/__init__.py:9``)."""
from .functions import (MSDeformAttnFunction, ms_deform_attn_backward, ms_deform_attn_core_pytorch,  # noqa: F401
                        ms_deform_attn_forward)
from .modules import MSDeformAttn  # noqa: F401
