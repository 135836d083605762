"""Model API of the reference (``models/__init__.py:16-17``)."""
from .lwdetr import LWDETR, PostProcess, build  # noqa: F401
from .nested import NestedTensor, nested_tensor_from_tensor_list  # noqa: F401


def build_model(args):
    return build(args)
