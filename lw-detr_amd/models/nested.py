"""Input container of the forward path: batched image tensor + padding mask.

Same contract as the reference's ``NestedTensor`` / ``nested_tensor_from_tensor_list``
(``util/misc.py:294-314, 317-339``): ``mask`` is True on padded pixels.
"""
from typing import List, Optional

import torch
from torch import Tensor


class NestedTensor:
    def __init__(self, tensors: Tensor, mask: Optional[Tensor]):
        self.tensors = tensors
        self.mask = mask

    def to(self, device):
        return NestedTensor(self.tensors.to(device), None if self.mask is None else self.mask.to(device))

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def nested_tensor_from_tensor_list(tensor_list) -> NestedTensor:
    """Pads a list of (3, h, w) images (or takes a (B,3,H,W) batch) to the batch maximum; mask marks the padding."""
    if isinstance(tensor_list, Tensor):
        if tensor_list.ndim != 4:
            raise ValueError("not supported")
        b, _, h, w = tensor_list.shape
        return NestedTensor(tensor_list, torch.zeros((b, h, w), dtype=torch.bool, device=tensor_list.device))
    if tensor_list[0].ndim != 3:
        raise ValueError("not supported")
    c = tensor_list[0].shape[0]
    h = max(img.shape[1] for img in tensor_list)
    w = max(img.shape[2] for img in tensor_list)
    dtype, device = tensor_list[0].dtype, tensor_list[0].device
    tensor = torch.zeros((len(tensor_list), c, h, w), dtype=dtype, device=device)
    mask = torch.ones((len(tensor_list), h, w), dtype=torch.bool, device=device)
    for img, pad_img, m in zip(tensor_list, tensor, mask):
        pad_img[:, : img.shape[1], : img.shape[2]].copy_(img)
        m[: img.shape[1], : img.shape[2]] = False
    return NestedTensor(tensor, mask)
