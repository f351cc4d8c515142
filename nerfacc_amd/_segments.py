"""Segment bookkeeping shared by the packed ops: a validated ``packed_info`` plus the tile
ownership table the flat segmented kernels need (csrc/segscan.hip).

The reference recomputes ``pack_info(ray_indices)`` inside every ``render_*`` call
(volrend.py:200-201, 256-257) and pays a device sync per boolean-index.  Here the derived data
is computed once per tensor and cached ON the tensor object (keyed by its version counter), and
tensors produced by this package (``sampling``, ``traverse_grids``, ``pack_info``) are
pre-tagged so that no validation read-back is needed for them.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor

from . import _backend as B


@dataclass
class SegInfo:
    packed_info: Tensor          # int64 [n_rays, 2], contiguous
    n_rays: int
    n_elems: int
    tiles: Optional[Tensor]      # int64 [nfa_seg_table_rows(n_tiles), 2]: {first ray, first element} per tile + the end sentinel; None when chunks are not contiguous
    contiguous: bool             # starts[r+1] == starts[r] + cnts[r]  (flat kernels usable)
    sorted_indices: bool = True  # for infos derived from ray_indices
    n_tiles: int = 0


_ATTR = "_nfa_seg"
_ATTR_RI = "_nfa_seg_from_ray_indices"


def _build_tiles(packed_info: Tensor, n_elems: int, trusted: bool) -> SegInfo:
    dev = B.require_device(packed_info)
    n_rays = packed_info.shape[0]
    with torch.cuda.device(dev):
        tile_elems, n_tiles = B.seg_plan(n_elems, n_rays)
        tiles = torch.empty((int(B.load().nfa_seg_table_rows(n_tiles)), 2), dtype=torch.int64, device=dev)
        flag = None if trusted else torch.empty(1, dtype=torch.int32, device=dev)  # (no flag: no memset launch either)
        B.call("nfa_seg_build_tiles", B.ptr(packed_info), n_rays, n_elems, tile_elems, n_tiles, B.ptr(tiles), B.ptr(flag),
               B.stream())
        ok = True if trusted else (int(flag.item()) == 0)  # one read-back for foreign packed_info
    return SegInfo(packed_info, n_rays, n_elems, tiles if ok else None, ok, n_tiles=n_tiles)


def seginfo_from_packed(packed_info: Tensor, n_elems: int, trusted: bool = False) -> SegInfo:
    """SegInfo for a user-visible ``packed_info`` tensor of shape (n_rays, 2)."""
    assert packed_info.dim() == 2 and packed_info.shape[-1] == 2, "packed_info must be 2-D with shape (B, 2)."
    cached = getattr(packed_info, _ATTR, None)
    if cached is not None and cached[0] == packed_info._version and cached[1] == n_elems:
        return cached[2]
    pi = packed_info
    if pi.dtype != torch.int64 or not pi.is_contiguous():
        pi = pi.to(torch.int64).contiguous()
    info = _build_tiles(pi, n_elems, trusted)
    try:
        setattr(packed_info, _ATTR, (packed_info._version, n_elems, info))
    except Exception:  # pragma: no cover - tensors always accept attributes; be defensive
        pass
    return info


def tag_trusted(packed_info: Tensor, n_elems: int) -> SegInfo:
    """Attach a SegInfo to a packed_info this package just produced (known contiguous)."""
    return seginfo_from_packed(packed_info, n_elems, trusted=True)


def pack_info_native(ray_indices: Tensor, n_rays: int):
    """(packed_info, unsorted_flag_tensor) via csrc/grid.hip (ref: pack.py:38-46)."""
    dev = B.require_device(ray_indices)
    ri = ray_indices if (ray_indices.dtype == torch.int64 and ray_indices.is_contiguous()) else \
        ray_indices.to(torch.int64).contiguous()
    with torch.cuda.device(dev):
        packed = torch.empty((n_rays, 2), dtype=torch.int64, device=dev)
        flag = torch.empty(1, dtype=torch.int32, device=dev)
        scratch = B.cumsum_scratch(n_rays, dev)
        B.call("nfa_pack_info", B.ptr(ri), ri.numel(), n_rays, B.ptr(packed), B.ptr(flag), B.ptr(scratch), B.stream())
    return packed, flag


def seginfo_from_ray_indices(ray_indices: Tensor, n_rays: int) -> SegInfo:
    """SegInfo derived from sample->ray indices, cached on the ``ray_indices`` tensor."""
    assert ray_indices.dim() == 1, "ray_indices must be a 1D tensor with shape (n_samples)."
    assert n_rays is not None, "n_rays must be provided"
    cached = getattr(ray_indices, _ATTR_RI, None)
    if cached is not None and cached[0] == ray_indices._version and cached[1] == n_rays:
        return cached[2]
    packed, flag = pack_info_native(ray_indices, n_rays)
    info = _build_tiles(packed, ray_indices.numel(), trusted=True)
    info.sorted_indices = int(flag.item()) == 0  # one read-back for foreign ray_indices
    try:
        setattr(ray_indices, _ATTR_RI, (ray_indices._version, n_rays, info))
    except Exception:  # pragma: no cover
        pass
    return info


def tag_ray_indices(ray_indices: Tensor, n_rays: int, info: SegInfo) -> None:
    """Pre-tag ray_indices produced by this package (sorted by construction)."""
    setattr(ray_indices, _ATTR_RI, (ray_indices._version, n_rays, info))


def resolve(n_elems: int, packed_info: Optional[Tensor], ray_indices: Optional[Tensor],
            n_rays: Optional[int]) -> Optional[SegInfo]:
    """The reference's rule (volrend.py:200-201): packed_info wins, else pack ray_indices."""
    if packed_info is not None:
        return seginfo_from_packed(packed_info, n_elems)
    if ray_indices is not None:
        if n_rays is None:
            n_rays = int(ray_indices.max().item()) + 1 if ray_indices.numel() else 0
        return seginfo_from_ray_indices(ray_indices, n_rays)
    return None


# ----------------------------------------------------------------------------- batched tensors
_UNIFORM: "dict[tuple, SegInfo]" = {}


def uniform_seginfo(n_rows: int, row_len: int, device: torch.device) -> SegInfo:
    """SegInfo of a batched ``(..., S)`` tensor viewed flat: ``n_rows`` rays of exactly ``row_len`` samples.

    The reference runs batched inputs through ``torch.cumsum`` / ``cumprod`` along the last dim (scan.py:42-44);
    on this GPU that kernel needs 2.7 ms for a (2^20, 64) tensor, the flat segmented engine 0.1 ms, so batched
    CUDA tensors take the same native path as packed ones.  A few shapes are cached (16 B per row).
    """
    key = (int(n_rows), int(row_len), device.type, device.index)
    info = _UNIFORM.get(key)
    if info is None:
        starts = torch.arange(n_rows, dtype=torch.int64, device=device) * row_len
        packed = torch.stack([starts, torch.full_like(starts, row_len)], dim=-1)
        info = _build_tiles(packed, n_rows * row_len, trusted=True)
        setattr(packed, _ATTR, (packed._version, n_rows * row_len, info))  # resolve() finds it without a read-back
        if len(_UNIFORM) >= 6:
            _UNIFORM.pop(next(iter(_UNIFORM)))
        _UNIFORM[key] = info
    return info


def batched_native(*tensors: Optional[Tensor]) -> Optional[SegInfo]:
    """Uniform SegInfo if the batched tensors (same shape, CUDA, float32, last dim = samples) can take the
    native path, else None (CPU tensors and other dtypes keep the reference's torch composition)."""
    ref = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() < 1:
            return None
        if ref is None:
            ref = t
        elif t.shape != ref.shape:
            return None
    if ref is None or ref.numel() == 0:
        return None
    return uniform_seginfo(ref.numel() // ref.shape[-1], ref.shape[-1], ref.device)
