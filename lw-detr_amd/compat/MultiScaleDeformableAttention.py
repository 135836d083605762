"""Drop-in for the reference's compiled extension module ``MultiScaleDeformableAttention``
(``models/ops/src/vision.cpp:13-16``): put ``lw-detr_amd/compat`` on ``sys.path`` and the reference's unmodified
``models/ops/functions/ms_deform_attn_func.py:23`` (``import MultiScaleDeformableAttention as MSDA``) binds to the
gfx950 kernel. See INTEGRATION.md."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _root not in sys.path:
    sys.path.insert(0, _root)

from lwdetr_amd.ops.functions import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: E402,F401
