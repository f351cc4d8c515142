"""``RaySamples`` / ``RayIntervals`` containers (ref: nerfacc/data_specs.py:12-180).

Same public fields.  The reference marshals these to a pybind ``RaySegmentsSpec``
(`_to_cpp` / `_from_cpp`); here the native layer takes plain pointers, so the marshalling is a
dictionary of tensors (``_to_spec``).  ``RaySamples._to_cpp`` in the reference reads a
non-existent ``self.chunk_cnts`` (data_specs.py:57) and always raises; ``_to_spec`` does what was
meant.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch


def _spec(vals, packed_info, ray_indices):
    spec = {"vals": vals.contiguous(), "packed_info": None, "ray_indices": None}
    if packed_info is not None:
        spec["packed_info"] = packed_info.to(torch.int64).contiguous()
    if ray_indices is not None:
        spec["ray_indices"] = ray_indices.to(torch.int64).contiguous()
    return spec


@dataclass
class RaySamples:
    """Ray samples, batched ``(n_rays, n_samples)`` or flattened ``(all_samples,)``.

    When ``vals`` is flattened either ``packed_info`` or ``ray_indices`` must be provided.
    """

    vals: torch.Tensor
    packed_info: Optional[torch.Tensor] = None
    ray_indices: Optional[torch.Tensor] = None
    is_valid: Optional[torch.Tensor] = None

    def _to_spec(self):
        return _spec(self.vals, self.packed_info, self.ray_indices)

    @property
    def device(self) -> torch.device:
        return self.vals.device


@dataclass
class RayIntervals:
    """Ray intervals: ``vals`` holds interval edges; ``is_left`` / ``is_right`` mark, for
    flattened data, whether an edge opens / closes an interval."""

    vals: torch.Tensor
    packed_info: Optional[torch.Tensor] = None
    ray_indices: Optional[torch.Tensor] = None
    is_left: Optional[torch.Tensor] = None
    is_right: Optional[torch.Tensor] = None

    def _to_spec(self):
        return _spec(self.vals, self.packed_info, self.ray_indices)

    @property
    def device(self) -> torch.device:
        return self.vals.device
