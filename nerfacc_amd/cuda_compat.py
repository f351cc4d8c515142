"""Drop-in replacement for the reference's native module ``_C``.

The reference resolves every native call lazily through ``getattr(_C, name)``
(nerfacc/cuda/__init__.py:8-15), where ``_C`` is the pybind11 module registered in
nerfacc/cuda/csrc/nerfacc.cpp:100-129.  This module exposes the same names with the same
positional signatures and return types, implemented on libnerfacc_hip.so (C ABI:
include/nerfacc_hip.h).  A maintainer of the reference switches backends with one line in
``nerfacc/cuda/_backend.py``::

    import nerfacc_amd.cuda_compat as _C

(see INTEGRATION.md).  ``RaySegmentsSpec`` mirrors the pybind class (read/write tensor attributes,
unset ones read as None).
"""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import torch
from torch import Tensor

from . import _backend as B
from . import grid as _grid
from . import pdf as _pdf
from .data_specs import RayIntervals
from .scan import _packed_scan_raw
from ._segments import seginfo_from_packed

__all__ = [
    "RaySegmentsSpec", "inclusive_sum", "exclusive_sum", "inclusive_prod_forward", "inclusive_prod_backward",
    "exclusive_prod_forward", "exclusive_prod_backward", "ray_aabb_intersect", "traverse_grids",
    "importance_sampling", "searchsorted", "opencv_lens_undistortion", "opencv_lens_undistortion_fisheye",
]


class RaySegmentsSpec:
    """ref: include/data_spec.hpp:6-107 + the pybind class at nerfacc.cpp:120-128."""

    def __init__(self) -> None:
        self.vals: Optional[Tensor] = None
        self.is_left: Optional[Tensor] = None
        self.is_right: Optional[Tensor] = None
        self.is_valid: Optional[Tensor] = None
        self.chunk_starts: Optional[Tensor] = None
        self.chunk_cnts: Optional[Tensor] = None
        self.ray_indices: Optional[Tensor] = None


def _check(t: Tensor, name: str) -> None:
    # CHECK_INPUT (include/utils_cuda.cuh:13-18)
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _packed(chunk_starts: Tensor, chunk_cnts: Tensor) -> Tensor:
    _check(chunk_starts, "chunk_starts")
    _check(chunk_cnts, "chunk_cnts")
    if chunk_starts.dim() != 1 or chunk_cnts.dim() != 1 or chunk_starts.shape != chunk_cnts.shape:
        raise RuntimeError("chunk_starts and chunk_cnts must be 1-D with equal sizes")
    return torch.stack([chunk_starts.to(torch.int64), chunk_cnts.to(torch.int64)], dim=-1)


def _scan(kind: int, chunk_starts, chunk_cnts, inputs, normalize: bool, backward: bool) -> Tensor:
    _check(inputs, "inputs")
    if inputs.dim() != 1:
        raise RuntimeError("inputs must be 1-D")
    seg = seginfo_from_packed(_packed(chunk_starts, chunk_cnts), inputs.numel())
    return _packed_scan_raw(kind, bool(backward), seg, inputs, bool(normalize))


def inclusive_sum(chunk_starts: Tensor, chunk_cnts: Tensor, inputs: Tensor, normalize: bool, backward: bool) -> Tensor:
    """ref: scan.cu:9-66."""
    return _scan(0, chunk_starts, chunk_cnts, inputs, normalize, backward)


def exclusive_sum(chunk_starts: Tensor, chunk_cnts: Tensor, inputs: Tensor, normalize: bool, backward: bool) -> Tensor:
    """ref: scan.cu:68-125."""
    return _scan(1, chunk_starts, chunk_cnts, inputs, normalize, backward)


def inclusive_prod_forward(chunk_starts: Tensor, chunk_cnts: Tensor, inputs: Tensor) -> Tensor:
    """ref: scan.cu:127-165."""
    return _scan(2, chunk_starts, chunk_cnts, inputs, False, False)


def exclusive_prod_forward(chunk_starts: Tensor, chunk_cnts: Tensor, inputs: Tensor) -> Tensor:
    """ref: scan.cu:217-257."""
    return _scan(3, chunk_starts, chunk_cnts, inputs, False, False)


def _prod_backward(kind: int, chunk_starts, chunk_cnts, inputs, outputs, grad_outputs) -> Tensor:
    _check(grad_outputs, "grad_outputs")
    seg = seginfo_from_packed(_packed(chunk_starts, chunk_cnts), inputs.numel())
    grad_outputs = grad_outputs.contiguous()
    if grad_outputs.numel() == 0:
        return torch.empty_like(grad_outputs)
    if not seg.contiguous:
        return _packed_scan_raw(kind - 2, True, seg, grad_outputs * outputs) / inputs.clamp_min(1e-10)
    grad_inputs = torch.empty_like(grad_outputs)
    with torch.cuda.device(grad_outputs.device):
        B.call("nfa_packed_prod_backward", kind, B.ptr(seg.packed_info), B.ptr(seg.tiles), seg.n_tiles, seg.n_rays,
               inputs.numel(), B.ptr(inputs.contiguous()), B.ptr(outputs.contiguous()), B.ptr(grad_outputs),
               B.ptr(grad_inputs), B.stream())
    return grad_inputs


def inclusive_prod_backward(chunk_starts, chunk_cnts, inputs, outputs, grad_outputs) -> Tensor:
    """ref: scan.cu:169-214."""
    return _prod_backward(2, chunk_starts, chunk_cnts, inputs, outputs, grad_outputs)


def exclusive_prod_backward(chunk_starts, chunk_cnts, inputs, outputs, grad_outputs) -> Tensor:
    """ref: scan.cu:259-304."""
    return _prod_backward(3, chunk_starts, chunk_cnts, inputs, outputs, grad_outputs)


def ray_aabb_intersect(rays_o: Tensor, rays_d: Tensor, aabbs: Tensor, near_plane: float, far_plane: float,
                       miss_value: float) -> List[Tensor]:
    """ref: grid.cu:477-519 -> [t_mins, t_maxs, hits]."""
    return list(_grid.ray_aabb_intersect(rays_o, rays_d, aabbs, near_plane, far_plane, miss_value))


def traverse_grids(rays_o, rays_d, rays_mask, binaries, aabbs, t_sorted, t_indices, hits, near_planes, far_planes,
                   step_size: float, cone_angle: float, compute_intervals: bool, compute_samples: bool,
                   compute_terminate_planes: bool, traverse_steps_limit: int, over_allocate: bool
                   ) -> Tuple[RaySegmentsSpec, RaySegmentsSpec, Optional[Tensor]]:
    """ref: grid.cu:320-474 (17 positional arguments, nerfacc.cpp:49-70)."""
    if over_allocate and traverse_steps_limit <= 0:
        raise RuntimeError("traverse_steps_limit must be > 0 when over_allocate is true")  # grid.cu:345
    iv, sm, term = _grid.traverse_grids(
        rays_o, rays_d, binaries, aabbs, near_planes=near_planes, far_planes=far_planes, step_size=step_size,
        cone_angle=cone_angle, traverse_steps_limit=traverse_steps_limit if traverse_steps_limit > 0 else None,
        over_allocate=over_allocate, rays_mask=rays_mask, t_sorted=t_sorted, t_indices=t_indices, hits=hits)
    intervals, samples = RaySegmentsSpec(), RaySegmentsSpec()
    if compute_intervals:
        intervals.vals, intervals.ray_indices = iv.vals, iv.ray_indices
        intervals.is_left, intervals.is_right = iv.is_left, iv.is_right
        intervals.chunk_starts, intervals.chunk_cnts = iv.packed_info[:, 0].contiguous(), iv.packed_info[:, 1].contiguous()
    if compute_samples:
        samples.vals, samples.ray_indices, samples.is_valid = sm.vals, sm.ray_indices, sm.is_valid
        samples.chunk_starts, samples.chunk_cnts = sm.packed_info[:, 0].contiguous(), sm.packed_info[:, 1].contiguous()
    return intervals, samples, (term if compute_terminate_planes else None)


def _intervals_from_spec(spec: RaySegmentsSpec) -> RayIntervals:
    if spec.vals is None:
        raise RuntimeError("RaySegmentsSpec.vals is undefined")  # data_spec.hpp:16-18
    pi = None
    if spec.vals.dim() == 1:
        if spec.chunk_starts is None or spec.chunk_cnts is None:
            raise RuntimeError("flattened RaySegmentsSpec needs chunk_starts and chunk_cnts")  # data_spec.hpp:24-31
        pi = torch.stack([spec.chunk_starts, spec.chunk_cnts], dim=-1)
    return RayIntervals(vals=spec.vals, packed_info=pi, ray_indices=spec.ray_indices, is_left=spec.is_left,
                        is_right=spec.is_right)


def importance_sampling(ray_segments: RaySegmentsSpec, cdfs: Tensor, n_intervels_per_ray: Union[Tensor, int],
                        stratified: bool) -> List[RaySegmentsSpec]:
    """ref: pdf.cu:294-421 (both overloads) -> [intervals, samples]."""
    out_iv, out_sm = _pdf.importance_sampling(_intervals_from_spec(ray_segments), cdfs, n_intervels_per_ray, stratified)
    iv, sm = RaySegmentsSpec(), RaySegmentsSpec()
    iv.vals, sm.vals = out_iv.vals, out_sm.vals
    if out_iv.packed_info is not None:   # Tensor counts: packed outputs (chunk_starts / chunk_cnts, ray_indices, masks)
        iv.chunk_starts, iv.chunk_cnts = out_iv.packed_info[:, 0].contiguous(), out_iv.packed_info[:, 1].contiguous()
        sm.chunk_starts, sm.chunk_cnts = out_sm.packed_info[:, 0].contiguous(), out_sm.packed_info[:, 1].contiguous()
        iv.ray_indices, sm.ray_indices = out_iv.ray_indices, out_sm.ray_indices
        iv.is_left, iv.is_right = out_iv.is_left, out_iv.is_right
    return [iv, sm]


def searchsorted(query: RaySegmentsSpec, key: RaySegmentsSpec) -> List[Tensor]:
    """ref: pdf.cu:426-456 -> [ids_left, ids_right]."""
    return list(_pdf.searchsorted(_intervals_from_spec(key), _intervals_from_spec(query)))


def opencv_lens_undistortion(*args, **kwargs):
    raise NotImplementedError("camera undistortion is outside the hot path this package accelerates (SURVEY.md 8)")


def opencv_lens_undistortion_fisheye(*args, **kwargs):
    raise NotImplementedError("camera undistortion is outside the hot path this package accelerates (SURVEY.md 8)")
