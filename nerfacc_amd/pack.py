"""``pack_info`` (ref: nerfacc/pack.py:10-49)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from ._segments import pack_info_native, tag_trusted


@torch.no_grad()
def pack_info(ray_indices: Tensor, n_rays: Optional[int] = None) -> Tensor:
    """Pack ``ray_indices`` to ``packed_info`` = (start, count) per ray, LongTensor (n_rays, 2).

    >>> pack_info(tensor([0, 0, 1, 1, 1, 2, 2, 2, 2]), n_rays=3)
    tensor([[0, 2], [2, 3], [5, 4]])

    Like the reference this needs a device tensor (pack.py:47-48 raises on CPU).  The histogram
    uses one atomic per run of equal indices instead of ``index_add_`` per sample.
    """
    assert ray_indices.dim() == 1, "ray_indices must be a 1D tensor with shape (n_samples)."
    if not ray_indices.is_cuda:
        raise NotImplementedError("Only support cuda inputs.")
    if n_rays is None:
        n_rays = int(ray_indices.max().item()) + 1 if ray_indices.numel() else 0
    packed, _ = pack_info_native(ray_indices, n_rays)
    tag_trusted(packed, ray_indices.numel())
    if ray_indices.dtype != torch.int64:
        packed = packed.to(ray_indices.dtype)  # the reference keeps the dtype of ray_indices
    return packed
