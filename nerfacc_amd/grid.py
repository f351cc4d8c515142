"""Ray/AABB intersection and occupancy-grid traversal (ref: nerfacc/grid.py).

``ray_aabb_intersect`` and ``traverse_grids`` keep the reference's signatures and return types
(grid.py:13-51, :93-192).  The host orchestration the reference does in C++
(cuda/csrc/grid.cu:320-474: count pass, cumsum, ``.item()``, allocation, fill pass) lives here
in Python over the C ABI; it needs ONE device->host read (both totals at once) instead of the
reference's two.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _backend as B
from ._segments import tag_ray_indices, tag_trusted
from .data_specs import RayIntervals, RaySamples


@torch.no_grad()
def ray_aabb_intersect(
    rays_o: Tensor,
    rays_d: Tensor,
    aabbs: Tensor,
    near_plane: float = -float("inf"),
    far_plane: float = float("inf"),
    miss_value: float = float("inf"),
) -> Tuple[Tensor, Tensor, Tensor]:
    """Ray-AABB slab test -> (t_mins (n_rays, m), t_maxs (n_rays, m), hits bool (n_rays, m)).

    Same contract as the reference (grid.py:13-51); kernel: csrc/grid.hip.
    """
    assert rays_o.ndim == 2 and rays_o.shape[-1] == 3
    assert rays_d.ndim == 2 and rays_d.shape[-1] == 3
    assert aabbs.ndim == 2 and aabbs.shape[-1] == 6
    dev = B.require_device(rays_o, rays_d, aabbs)
    rays_o, rays_d, aabbs = rays_o.float().contiguous(), rays_d.float().contiguous(), aabbs.float().contiguous()
    n, m = rays_o.shape[0], aabbs.shape[0]
    t_mins = torch.empty((n, m), dtype=torch.float32, device=dev)
    t_maxs = torch.empty((n, m), dtype=torch.float32, device=dev)
    hits = torch.empty((n, m), dtype=torch.bool, device=dev)
    with torch.cuda.device(dev):
        B.call("nfa_ray_aabb_intersect", B.ptr(rays_o), B.ptr(rays_d), n, B.ptr(aabbs), m, float(near_plane),
               float(far_plane), float(miss_value), B.ptr(t_mins), B.ptr(t_maxs), B.ptr(hits), B.stream())
    return t_mins, t_maxs, hits


def _traverse_args(rays_o, rays_d, rays_mask, binaries, aabbs, t_sorted, t_indices, hits, near_planes, far_planes,
                   step_size, cone_angle, limit, mode) -> B.TraverseArgs:
    a = B.TraverseArgs()
    a.n_rays = rays_o.shape[0]
    a.rays_o, a.rays_d, a.rays_mask = B.ptr(rays_o), B.ptr(rays_d), B.ptr(rays_mask)
    a.n_grids = binaries.shape[0]
    a.res[0], a.res[1], a.res[2] = binaries.shape[1], binaries.shape[2], binaries.shape[3]
    a.binaries, a.aabbs = B.ptr(binaries), B.ptr(aabbs)
    a.hits, a.t_sorted, a.t_indices = B.ptr(hits), B.ptr(t_sorted), B.ptr(t_indices)
    a.near_planes, a.far_planes = B.ptr(near_planes), B.ptr(far_planes)
    a.step_size, a.cone_angle = float(step_size), float(cone_angle)
    a.traverse_steps_limit, a.mode = int(limit), int(mode)
    bricks, coarse = _get_bricks(binaries)  # cached on the tensor
    a._keepalive = (bricks, coarse)
    a.bricks, a.coarse = B.ptr(bricks), B.ptr(coarse)
    return a


def _launch(a: B.TraverseArgs) -> None:
    B.call("nfa_traverse_grids", C.byref(a), B.stream())


def _exclusive_cumsum(cnts: Tensor, total_out: Tensor):
    starts = torch.empty_like(cnts)
    scratch = B.cumsum_scratch(cnts.numel(), cnts.device)
    B.call("nfa_exclusive_cumsum_i64", B.ptr(cnts), cnts.numel(), B.ptr(starts), B.ptr(total_out), B.ptr(scratch),
           B.stream())
    return starts


def _cumsum_packed(cnts: Tensor, total_out: Tensor, stats: bool = False) -> Tensor:
    """packed_info rows {exclusive start, count} from per-ray counts (one pass, no torch.stack).  With ``stats``,
    ``total_out`` has three slots: the total and the two sums of the coherence measure (include/nerfacc_hip.h)."""
    packed = torch.empty((cnts.numel(), 2), dtype=torch.int64, device=cnts.device)
    scratch = B.cumsum_scratch(cnts.numel(), cnts.device)
    B.call("nfa_exclusive_cumsum_pairs_stats_i64" if stats else "nfa_exclusive_cumsum_pairs_i64", B.ptr(cnts), cnts.numel(),
           B.ptr(packed), B.ptr(total_out), B.ptr(scratch), B.stream())
    return packed


MAX_EVENT_LEVELS = 8   # include/nerfacc_hip.h: NFA_MAX_EVENT_LEVELS


def ray_events(rays_o: Tensor, rays_d: Tensor, aabbs: Tensor):
    """``(t_sorted, t_indices, hits)`` of the reference's ``traverse_grids`` preamble (grid.py:156-162:
    ``ray_aabb_intersect`` with its defaults, then ``torch.sort(torch.cat([t_mins, t_maxs], -1), -1)``) in one native
    pass (``nfa_ray_events``: intersection and a stable sort of the 2 G distances in registers) for up to 8 nested boxes;
    more than that: the reference's composition on the native intersection."""
    G = aabbs.shape[0]
    if G > MAX_EVENT_LEVELS or not rays_o.is_cuda:
        t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, aabbs)
        t_sorted, t_indices = torch.sort(torch.cat([t_mins, t_maxs], dim=-1), dim=-1)
        return t_sorted, t_indices, hits
    dev = B.require_device(rays_o, rays_d, aabbs)
    rays_o, rays_d, aabbs = rays_o.float().contiguous(), rays_d.float().contiguous(), aabbs.float().contiguous()
    n = rays_o.shape[0]
    t_sorted = torch.empty((n, 2 * G), dtype=torch.float32, device=dev)
    t_indices = torch.empty((n, 2 * G), dtype=torch.int64, device=dev)
    hits = torch.empty((n, G), dtype=torch.bool, device=dev)
    with torch.cuda.device(dev):
        B.call("nfa_ray_events", B.ptr(rays_o), B.ptr(rays_d), n, B.ptr(aabbs), G, B.ptr(t_sorted), B.ptr(t_indices),
               B.ptr(hits), B.stream())
    return t_sorted, t_indices, hits


def _prepare(rays_o, rays_d, binaries, aabbs, near_planes, far_planes, rays_mask, t_sorted, t_indices, hits,
             allow_fused: bool):
    dev = B.require_device(rays_o, rays_d, binaries, aabbs)
    assert rays_o.ndim == 2 and rays_o.shape[-1] == 3 and rays_d.shape == rays_o.shape
    assert binaries.ndim == 4 and aabbs.ndim == 2 and aabbs.shape == (binaries.shape[0], 6)
    rays_o, rays_d, aabbs = rays_o.float().contiguous(), rays_d.float().contiguous(), aabbs.float().contiguous()
    binaries = (binaries if binaries.dtype == torch.bool else binaries.bool()).contiguous()
    n_rays = rays_o.shape[0]
    if near_planes is None:
        near_planes = torch.zeros(n_rays, dtype=torch.float32, device=dev)
    if far_planes is None:
        far_planes = torch.full((n_rays,), float("inf"), dtype=torch.float32, device=dev)
    near_planes, far_planes = near_planes.float().contiguous(), far_planes.float().contiguous()
    if rays_mask is not None:
        rays_mask = rays_mask.bool().contiguous()
    if t_sorted is None or t_indices is None or hits is None:
        if allow_fused and binaries.shape[0] == 1:
            t_sorted = t_indices = hits = None  # intersected inside the traversal kernel
        else:  # grid.py:156-162
            t_sorted, t_indices, hits = ray_events(rays_o, rays_d, aabbs)
    if t_sorted is not None:
        t_sorted = t_sorted.float().contiguous()
        t_indices = t_indices.to(torch.int64).contiguous()
        hits = hits.bool().contiguous()
    return dev, rays_o, rays_d, binaries, aabbs, near_planes, far_planes, rays_mask, t_sorted, t_indices, hits


@torch.no_grad()
def traverse_grids(
    rays_o: Tensor,  # [n_rays, 3]
    rays_d: Tensor,  # [n_rays, 3]
    binaries: Tensor,  # [m, resx, resy, resz]
    aabbs: Tensor,  # [m, 6]
    near_planes: Optional[Tensor] = None,  # [n_rays]
    far_planes: Optional[Tensor] = None,  # [n_rays]
    step_size: Optional[float] = 1e-3,
    cone_angle: Optional[float] = 0.0,
    traverse_steps_limit: Optional[int] = None,
    over_allocate: Optional[bool] = False,
    rays_mask: Optional[Tensor] = None,  # [n_rays]
    t_sorted: Optional[Tensor] = None,  # [n_rays, n_grids * 2]
    t_indices: Optional[Tensor] = None,  # [n_rays, n_grids * 2]
    hits: Optional[Tensor] = None,  # [n_rays, n_grids]
) -> Tuple[RayIntervals, RaySamples, Tensor]:
    """Ray traversal within multiple grids (not differentiable).

    Arguments and returns as the reference (grid.py:93-192): a :class:`RayIntervals`, a
    :class:`RaySamples` and the per-ray termination planes.  ``rays_mask`` is only honoured with
    ``over_allocate`` (grid.cu:418,450 pass nullptr in two-pass mode).  Entries of the
    termination planes the reference leaves uninitialised are defined here: the ray's near plane
    for masked rays, the count pass's value for rays without samples.
    """
    near_hint = 0.0 if near_planes is None else None  # the default near plane is a constant (accelerator only)
    if traverse_steps_limit is None:
        traverse_steps_limit = -1
    if over_allocate:
        assert traverse_steps_limit > 0, "traverse_steps_limit must be set if over_allocate is True."
    (dev, rays_o, rays_d, binaries, aabbs, near_planes, far_planes, rays_mask, t_sorted, t_indices,
     hits) = _prepare(rays_o, rays_d, binaries, aabbs, near_planes, far_planes, rays_mask, t_sorted, t_indices, hits,
                      allow_fused=True)
    n_rays = rays_o.shape[0]
    i64 = dict(dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        terminate = torch.empty(n_rays, dtype=torch.float32, device=dev)
        totals = torch.empty(2, **i64)
        iv_packed = sm_packed = None
        if over_allocate:  # grid.cu:364-404
            limit = int(traverse_steps_limit)
            mask_l = torch.ones(n_rays, **i64) if rays_mask is None else rays_mask.to(torch.int64)
            sm_alloc = mask_l * limit
            iv_alloc = sm_alloc * 2
            sm_off = torch.cumsum(sm_alloc, 0) - sm_alloc
            iv_off = sm_off * 2
            n_alive = int(mask_l.sum().item()) if rays_mask is not None else n_rays
            n_sm, n_iv = n_alive * limit, n_alive * limit * 2
            iv_cnts, sm_cnts = torch.empty(n_rays, **i64), torch.empty(n_rays, **i64)
            iv_vals = torch.zeros(n_iv, dtype=torch.float32, device=dev)
            iv_ri = torch.zeros(n_iv, **i64)
            iv_l = torch.zeros(n_iv, dtype=torch.bool, device=dev)
            iv_r = torch.zeros(n_iv, dtype=torch.bool, device=dev)
            sm_vals = torch.zeros(n_sm, dtype=torch.float32, device=dev)
            sm_ri = torch.zeros(n_sm, **i64)
            sm_valid = torch.zeros(n_sm, dtype=torch.bool, device=dev)
            a = _traverse_args(rays_o, rays_d, rays_mask, binaries, aabbs, t_sorted, t_indices, hits, near_planes,
                               far_planes, step_size, cone_angle, limit, 2)
            a.iv_vals, a.iv_ray_indices, a.iv_is_left, a.iv_is_right = B.ptr(iv_vals), B.ptr(iv_ri), B.ptr(iv_l), B.ptr(iv_r)
            a.iv_starts, a.iv_cnts = B.ptr(iv_off), B.ptr(iv_cnts)
            a.sm_vals, a.sm_ray_indices, a.sm_is_valid = B.ptr(sm_vals), B.ptr(sm_ri), B.ptr(sm_valid)
            a.sm_starts, a.sm_cnts = B.ptr(sm_off), B.ptr(sm_cnts)
            a.terminate_planes = B.ptr(terminate)
            _launch(a)
            # chunk_starts from the ACTUAL counts: the layout after the caller compacts with the
            # masks (grid.cu:401-403, examples/utils.py:362-365).
            iv_starts = _exclusive_cumsum(iv_cnts, totals[0:1])
            sm_starts = _exclusive_cumsum(sm_cnts, totals[1:2])
        elif float(step_size) > 0.0 and float(cone_angle) == 0.0 and _walk_supported(binaries):
            # constant step: ONE walk (run records) + two coalesced expansions instead of the reference's count and
            # fill passes (grid.cu:405-471); rays with more than MAX_RUNS runs are filled by the serial kernel.
            iv_cnts, sm_cnts = torch.empty(n_rays, **i64), torch.empty(n_rays, **i64)
            meta = torch.zeros(3, **i64)  # [edges, samples, rays with too many runs]
            a = _traverse_args(rays_o, rays_d, None, binaries, aabbs, t_sorted, t_indices, hits, near_planes,
                               far_planes, step_size, cone_angle, traverse_steps_limit, 0)
            a.iv_cnts, a.sm_cnts, a.terminate_planes = B.ptr(iv_cnts), B.ptr(sm_cnts), B.ptr(terminate)
            bits = _get_walk_bits(binaries)
            run_cnts = torch.empty(n_rays, dtype=torch.int32, device=dev)
            runs = torch.empty((MAX_RUNS, n_rays), dtype=torch.int64, device=dev)
            while True:
                B.call("nfa_traverse_runs", C.byref(a), B.ptr(bits), B.ptr(run_cnts), B.ptr(runs), MAX_RUNS,
                       B.ptr(meta[2:3]), float("nan") if near_hint is None else near_hint, None, 0, B.stream())
                iv_packed = _cumsum_packed(iv_cnts, meta[0:1])
                sm_packed = _cumsum_packed(sm_cnts, meta[1:2])
                n_iv, n_sm, n_overflow = (int(v) for v in meta.tolist())  # the one device->host read
                n_overflow, off_lattice = n_overflow & 0xFFFFFFFF, n_overflow >> 32   # (nfa_traverse_runs: [1] = off the lattice)
                if not off_lattice or near_hint is None:
                    break
                near_hint = None   # rays that are not on the lattice of near_hint (csrc/walk.hip): the per-ray marcher
                meta.zero_()
            iv_vals = torch.empty(n_iv, dtype=torch.float32, device=dev)
            iv_ri = torch.empty(n_iv, **i64)
            iv_l = torch.empty(n_iv, dtype=torch.bool, device=dev)
            iv_r = torch.empty(n_iv, dtype=torch.bool, device=dev)
            sm_vals = torch.empty(n_sm, dtype=torch.float32, device=dev)
            sm_ri = torch.empty(n_sm, **i64)
            sm_valid = torch.ones(n_sm, dtype=torch.bool, device=dev)
            if n_sm > 0:
                B.call("nfa_expand_runs", n_rays, float(step_size), B.ptr(run_cnts), B.ptr(runs), MAX_RUNS, B.ptr(sm_packed),
                       None, None, B.ptr(sm_vals), B.ptr(sm_ri), n_sm, B.stream())
                B.call("nfa_expand_intervals", n_rays, float(step_size), B.ptr(run_cnts), B.ptr(runs), MAX_RUNS,
                       B.ptr(iv_packed), B.ptr(iv_vals), B.ptr(iv_ri), B.ptr(iv_l), B.ptr(iv_r), B.stream())
                if n_overflow > 0:
                    iv_starts, sm_starts = iv_packed[:, 0].contiguous(), sm_packed[:, 0].contiguous()
                    a.mode = 1
                    a.terminate_planes = None
                    a.iv_vals, a.iv_ray_indices, a.iv_is_left, a.iv_is_right = (B.ptr(iv_vals), B.ptr(iv_ri), B.ptr(iv_l),
                                                                                 B.ptr(iv_r))
                    a.iv_starts = B.ptr(iv_starts)
                    a.sm_vals, a.sm_ray_indices, a.sm_is_valid = B.ptr(sm_vals), B.ptr(sm_ri), B.ptr(sm_valid)
                    a.sm_starts = B.ptr(sm_starts)
                    a.ray_filter, a.ray_filter_min = B.ptr(run_cnts), MAX_RUNS
                    _launch(a)
        else:  # two passes, grid.cu:405-471
            iv_cnts, sm_cnts = torch.empty(n_rays, **i64), torch.empty(n_rays, **i64)
            a = _traverse_args(rays_o, rays_d, None, binaries, aabbs, t_sorted, t_indices, hits, near_planes,
                               far_planes, step_size, cone_angle, traverse_steps_limit, 0)
            a.iv_cnts, a.sm_cnts, a.terminate_planes = B.ptr(iv_cnts), B.ptr(sm_cnts), B.ptr(terminate)
            _launch(a)
            iv_starts = _exclusive_cumsum(iv_cnts, totals[0:1])
            sm_starts = _exclusive_cumsum(sm_cnts, totals[1:2])
            n_iv, n_sm = (int(v) for v in totals.tolist())  # the one device->host read
            iv_vals = torch.empty(n_iv, dtype=torch.float32, device=dev)
            iv_ri = torch.empty(n_iv, **i64)
            iv_l = torch.empty(n_iv, dtype=torch.bool, device=dev)
            iv_r = torch.empty(n_iv, dtype=torch.bool, device=dev)
            sm_vals = torch.empty(n_sm, dtype=torch.float32, device=dev)
            sm_ri = torch.empty(n_sm, **i64)
            sm_valid = torch.empty(n_sm, dtype=torch.bool, device=dev)
            if n_sm > 0:
                a.mode = 1
                a.terminate_planes = None
                a.iv_vals, a.iv_ray_indices, a.iv_is_left, a.iv_is_right = B.ptr(iv_vals), B.ptr(iv_ri), B.ptr(iv_l), B.ptr(iv_r)
                a.iv_starts = B.ptr(iv_starts)
                a.sm_vals, a.sm_ray_indices, a.sm_is_valid = B.ptr(sm_vals), B.ptr(sm_ri), B.ptr(sm_valid)
                a.sm_starts = B.ptr(sm_starts)
                _launch(a)
        if iv_packed is None:
            iv_packed = torch.stack([iv_starts, iv_cnts], dim=-1)
            sm_packed = torch.stack([sm_starts, sm_cnts], dim=-1)
    if not over_allocate:
        tag_trusted(iv_packed, n_iv)
        info = tag_trusted(sm_packed, n_sm)
        tag_ray_indices(sm_ri, n_rays, info)
    intervals = RayIntervals(vals=iv_vals, packed_info=iv_packed, ray_indices=iv_ri, is_left=iv_l, is_right=iv_r)
    samples = RaySamples(vals=sm_vals, packed_info=sm_packed, ray_indices=sm_ri, is_valid=sm_valid)
    return intervals, samples, terminate


MAX_RUNS = int(os.environ.get("NERFACC_AMD_MAX_RUNS", "32"))  # runs kept per ray by the run-length traversal (csrc/walk.hip)


def _get_bricks(binaries: Tensor):
    """Brick-packed copy of ``binaries`` (derived cache, keyed on the tensor's version counter;
    never serialised -- the state_dict keeps the reference's torch.bool buffer)."""
    cached = getattr(binaries, "_nfa_bricks", None)
    if cached is not None and cached[0] == binaries._version:
        return cached[1], cached[2]
    dev = binaries.device
    res = (C.c_int32 * 3)(*binaries.shape[1:])
    words = int(B.load().nfa_bricks_words(binaries.shape[0], res))
    bricks = torch.empty(words, dtype=torch.int64, device=dev)
    coarse = torch.empty((words + 31) // 32, dtype=torch.int32, device=dev)
    B.call("nfa_pack_bricks", B.ptr(binaries), binaries.shape[0], res, B.ptr(bricks), B.ptr(coarse), B.stream())
    try:
        binaries._nfa_bricks = (binaries._version, bricks, coarse)
    except Exception:  # pragma: no cover
        pass
    return bricks, coarse


WALK_MAX_RES = 512  # cells per axis the run-length walk's packed step counters cover (csrc/walk.hip)


def _walk_supported(binaries: Tensor) -> bool:
    """The run-length walk's limits (csrc/walk.hip): at most 512 cells per axis, and the bit-interleaved cell index of all
    levels together below 2^31 (``n_grids << bits``; asked from the library, which owns the layout); grids beyond take
    the generic kernels of ``nfa_traverse_grids``."""
    if max(binaries.shape[1:]) > WALK_MAX_RES:
        return False
    res = (C.c_int32 * 3)(*binaries.shape[1:])
    return int(B.load().nfa_walk_bits_words(binaries.shape[0], res)) * 32 < (1 << 31)


def _get_walk_bits(binaries: Tensor) -> Tensor:
    """1-bit-per-cell copy of ``binaries`` in the run-length walk's blocked cell order (derived cache keyed on the
    tensor's version counter, never serialised)."""
    cached = getattr(binaries, "_nfa_walk_bits", None)
    if cached is not None and cached[0] == binaries._version:
        return cached[1]
    res = (C.c_int32 * 3)(*binaries.shape[1:])
    words = int(B.load().nfa_walk_bits_words(binaries.shape[0], res))
    bits = torch.empty(words, dtype=torch.int32, device=binaries.device)
    B.call("nfa_pack_walk_bits", B.ptr(binaries), binaries.shape[0], res, B.ptr(bits), B.stream())
    try:
        binaries._nfa_walk_bits = (binaries._version, bits)
    except Exception:  # pragma: no cover
        pass
    return bits


_SIDE_STREAMS: dict = {}


def _side_stream(dev) -> "torch.cuda.Stream":
    """A second stream per (device, current stream) for work that may run beside the caller's stream."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


SPECULATE = os.environ.get("NERFACC_AMD_SPECULATE", "1") != "0"   # 0: read the traversal's total before allocating its outputs
_SPEC_CAPACITY: dict = {}
_PINNED: dict = {}


def _pinned_meta(dev) -> Tensor:
    buf = _PINNED.get(dev.index)
    if buf is None:
        buf = _PINNED[dev.index] = torch.empty(8, dtype=torch.int64, pin_memory=True)
    return buf


CONE_RUNS = os.environ.get("NERFACC_AMD_CONE_RUNS", "1") != "0"   # 0: the serial count + fill passes (A/B testing)
ALIVE_LIST_FRACTION = float(os.environ.get("NERFACC_AMD_ALIVE_FRACTION", "0.75"))   # test-mode loop: below this share of alive rays only they are walked
CONE_WALK = os.environ.get("NERFACC_AMD_CONE_WALK", "1") != "0"   # 0: the count pass over the brick-packed grid (grid.hip) instead of walk.hip's DDA (A/B testing)
CONE_ARENA = os.environ.get("NERFACC_AMD_CONE_ARENA", "1") != "0"   # 0: rays with more than MAX_RUNS records go to the serial fill pass (A/B testing)
CONE_ARENA_MIN = 4096   # smallest arena (entries; a multiple of 16)
CONE_BIN_THRESHOLD = 1.25   # cone-angle walk: bin the rays when a wave of 64 neighbours crosses this many times its mean ray's cells


@torch.no_grad()
def _traverse_samples(rays_o, rays_d, binaries, aabbs, near_planes, far_planes, step_size, cone_angle,
                      rays_mask=None, traverse_steps_limit=None, t_sorted=None, t_indices=None, hits=None,
                      return_terminate=False, near_hint=None, bin_rays=False, stats_sink=None, speculate=True, n_alive=None,
                      alive_list=None):
    """Sampler fast path: (ray_indices, t_starts, t_ends, packed_info) straight from the traversal.

    Same values as ``intervals.vals[is_left]``, ``intervals.vals[is_right]``,
    ``samples.ray_indices``, ``samples.packed_info`` of :func:`traverse_grids`
    (ref: estimators/occ_grid.py:164-177) without materialising edges and masks and without
    the two boolean-index syncs.  Constant-step marching (``cone_angle == 0``) takes the
    run-length path: one DDA walk per ray + a fully parallel, coalesced expansion.

    With ``rays_mask`` / ``traverse_steps_limit`` (constant step, or a cone angle with a step limit: mask and limit are honoured
    inside the walk, which emits the compact arrays directly) this is the compacted equivalent of the reference's
    ``traverse_grids(over_allocate=True, rays_mask=..., traverse_steps_limit=...)`` followed by the
    ``is_left`` / ``is_right`` / ``is_valid`` boolean indexing of examples/utils.py:342-365: masked
    rays get no samples, every other ray at most ``traverse_steps_limit``.  ``n_alive``: the number of True entries of
    ``rays_mask`` when the caller knows it (the test-mode loop reads it anyway): with fewer than three quarters of the rays
    alive only those are walked, packed into dense waves -- a dead ray then costs no lane (it used to cost its wave the
    lane's slot for as long as the wave's longest alive ray walked).  ``alive_list``: the ids of those rays when the caller has
    them already (int32, at least ``n_alive`` entries: ``nfa_alive_rays``); used whatever the share.
    """
    limit = -1 if traverse_steps_limit is None else int(traverse_steps_limit)
    use_runs = float(step_size) > 0.0 and float(cone_angle) == 0.0 and _walk_supported(binaries)
    # distance-dependent steps: run records from the count pass + a coalesced expansion instead of a second walk
    use_cone_runs = CONE_RUNS and float(step_size) > 0.0 and float(cone_angle) > 0.0
    if (rays_mask is not None or limit > 0) and (not (use_runs or use_cone_runs) or (use_cone_runs and limit <= 0)):
        # marching with a cone angle / per-cell mode: the reference's route (over-allocate + compaction)
        iv, sm, term = traverse_grids(rays_o, rays_d, binaries, aabbs, near_planes, far_planes, step_size, cone_angle,
                                      limit if limit > 0 else None, limit > 0, rays_mask, t_sorted, t_indices, hits)
        valid = sm.is_valid if sm.is_valid is not None else torch.ones_like(sm.ray_indices, dtype=torch.bool)
        out = (sm.ray_indices[valid], iv.vals[iv.is_left], iv.vals[iv.is_right], sm.packed_info)
        return (*out, term) if return_terminate else out
    (dev, rays_o, rays_d, binaries, aabbs, near_planes, far_planes, rays_mask, t_sorted, t_indices,
     hits) = _prepare(rays_o, rays_d, binaries, aabbs, near_planes, far_planes, rays_mask, t_sorted, t_indices, hits,
                      allow_fused=True)
    n_rays = rays_o.shape[0]
    with torch.cuda.device(dev):
        masked = rays_mask is not None or limit > 0
        spec_key = (n_rays, dev.index)
        binned = bool(bin_rays) and n_rays >= 4096
        alive = None   # ids of the alive rays, when only they are walked
        if (alive_list is not None and n_alive is not None and rays_mask is not None and (use_runs or use_cone_runs) and not binned
                and 0 <= n_alive <= alive_list.numel() and alive_list.dtype == torch.int32 and alive_list.device == dev):
            alive = alive_list[:int(n_alive)]
            sm_cnts = torch.zeros(n_rays, dtype=torch.int64, device=dev)          # (the rays not listed keep these)
            terminate = near_planes.clone() if return_terminate else None
        elif (rays_mask is not None and n_alive is not None and (use_runs or use_cone_runs) and not binned
                and 0 <= n_alive and n_alive < ALIVE_LIST_FRACTION * n_rays):
            alive = torch.nonzero_static(rays_mask, size=int(n_alive)).view(-1).to(torch.int32)
            sm_cnts = torch.zeros(n_rays, dtype=torch.int64, device=dev)          # (the rays not listed keep these)
            terminate = near_planes.clone() if return_terminate else None
        else:
            sm_cnts = torch.empty(n_rays, dtype=torch.int64, device=dev)
            terminate = torch.empty(n_rays, dtype=torch.float32, device=dev) if return_terminate else None
        # [total samples, coherence sums (2), rays with too many runs, coherence sums of the cone walk's cell-count key (2)]
        meta = torch.zeros(6, dtype=torch.int64, device=dev)
        a = _traverse_args(rays_o, rays_d, rays_mask, binaries, aabbs, t_sorted, t_indices, hits, near_planes, far_planes,
                           step_size, cone_angle, limit, 2 if ((use_runs or use_cone_runs) and masked) else 0)
        a.sm_cnts = B.ptr(sm_cnts)
        a.terminate_planes = B.ptr(terminate)
        if use_runs:
            bits = _get_walk_bits(binaries)
            run_cnts = (torch.zeros if alive is not None else torch.empty)(n_rays, dtype=torch.int32, device=dev)
            runs = torch.empty((MAX_RUNS, n_rays), dtype=torch.int64, device=dev)  # slot-major run records
            order = alive
            if binned:
                # unrelated rays: lanes of a wave get rays of similar path length (results do not depend on it)
                order = torch.empty(n_rays, dtype=torch.int32, device=dev)
                scratch = torch.empty(1024 + n_rays, dtype=torch.uint8, device=dev)
                B.call("nfa_bin_rays", B.ptr(rays_o), B.ptr(rays_d), n_rays, B.ptr(aabbs[-1]), B.ptr(order), B.ptr(scratch),
                       B.stream())
            # near_hint: the scalar near plane when the caller built near_planes from one (accelerator only)
            B.call("nfa_traverse_runs", C.byref(a), B.ptr(bits), B.ptr(run_cnts), B.ptr(runs), MAX_RUNS,
                   B.ptr(meta[3:4]), float("nan") if near_hint is None else float(near_hint), B.ptr(order),
                   0 if order is None else order.numel(), B.stream())
        elif use_cone_runs:
            _get_bricks(binaries)
            arena, arena_cap = None, 0
            run_cnts = (torch.zeros if alive is not None else torch.empty)(n_rays, dtype=torch.int32, device=dev)
            runs = torch.empty((MAX_RUNS, n_rays), dtype=torch.int64, device=dev)
            order = alive
            if binned or (bin_rays is None and n_rays >= 65536 and alive is None and limit <= 0):
                # Unrelated rays: a wave runs as long as its longest ray, so its lanes get rays that cross about as many cells
                # (results do not depend on it).  bin_rays None: the key is computed anyway (0.06 ms per 2 M rays beside a walk
                # of milliseconds) for its coherence measure, and the assignment is used when the PREVIOUS batch's measure
                # said that neighbouring rays differ (image-ordered rays lose 8 % when binned, unrelated ones gain 10-17 %).
                # (Not for limited walks: their length is decided by the next occupied cells, and the kernel hands rays to
                # lanes as they become free.)
                order = torch.empty(n_rays, dtype=torch.int32, device=dev)
                scratch = torch.empty(1024 + n_rays, dtype=torch.uint8, device=dev)
                res3 = (C.c_int32 * 3)(*binaries.shape[1:])
                B.call("nfa_bin_rays_levels", B.ptr(rays_o), B.ptr(rays_d), n_rays, B.ptr(aabbs), binaries.shape[0], res3,
                       float(near_hint) if near_hint is not None else 0.0, B.ptr(order), B.ptr(scratch), B.ptr(meta[4:6]), B.stream())
                if not binned and not (stats_sink is not None and stats_sink.get("cells_max_over_mean", 1.0) > CONE_BIN_THRESHOLD):
                    order = None
            if CONE_WALK and _walk_supported(binaries) and min(binaries.shape[1:]) >= 4:
                # the constant-step walk's DDA over the 1-bit grid copy (csrc/walk.hip: cone_walk_kernel / cone_refill_kernel)
                # records beyond a ray's MAX_RUNS slots go to an arena (16 bytes each) instead of a second walk of the ray
                arena_cap = CONE_ARENA and max(CONE_ARENA_MIN, (n_rays // 8 + 15) // 16 * 16)
                arena = torch.zeros(2 * arena_cap, dtype=torch.int64, device=dev) if arena_cap else None
                B.call("nfa_traverse_cone_walk", C.byref(a), B.ptr(_get_walk_bits(binaries)), B.ptr(run_cnts), B.ptr(runs), MAX_RUNS,
                       B.ptr(meta[3:4]), B.ptr(arena), arena_cap, B.ptr(order), 0 if order is None else order.numel(), B.stream())
            else:
                B.call("nfa_traverse_cone_runs", C.byref(a), B.ptr(run_cnts), B.ptr(runs), MAX_RUNS, B.ptr(meta[3:4]), B.ptr(order),
                       0 if order is None else order.numel(), B.stream())
        else:
            _launch(a)
        packed_info = _cumsum_packed(sm_cnts, meta[0:3], stats=True)
        # The total is needed on the host (the outputs' shape).  With a capacity remembered from the previous batch of
        # this shape the constant-step expansion is launched BEFORE the host knows the total: the size travels on a side
        # stream that waits for the cumsum only, so the host reads it while the expansion runs and the launches that follow
        # queue up behind it -- no idle GPU between the read and the next kernel.  (A total above the capacity: the
        # expansion wrote nothing beyond it and is run again into arrays of the right size.)
        cap = _SPEC_CAPACITY.get(spec_key, 0) if (SPECULATE and speculate and use_runs) else 0
        t_starts = t_ends = ray_indices = None
        if cap > 0:
            host = _pinned_meta(dev)
            main = torch.cuda.current_stream()
            ready = torch.cuda.Event(); ready.record(main)
            side = _side_stream(dev)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                host[:6].copy_(meta, non_blocking=True)
                done = torch.cuda.Event(); done.record(side)
            t_starts = torch.empty(cap, dtype=torch.float32, device=dev)
            t_ends = torch.empty(cap, dtype=torch.float32, device=dev)
            ray_indices = torch.empty(cap, dtype=torch.int64, device=dev)
            B.call("nfa_expand_runs", n_rays, float(step_size), B.ptr(run_cnts), B.ptr(runs), MAX_RUNS,
                   B.ptr(packed_info), B.ptr(t_starts), B.ptr(t_ends), None, B.ptr(ray_indices), cap, B.stream())
            done.synchronize()
            n_sm, s_max, s_sum, n_overflow, c_max, c_sum = (int(v) for v in host[:6].tolist())
        else:
            n_sm, s_max, s_sum, n_overflow, c_max, c_sum = (int(v) for v in meta.tolist())  # the one device->host read of the traversal
        n_overflow, off_lattice = n_overflow & 0xFFFFFFFF, n_overflow >> 32   # (nfa_traverse_runs: [1] = off the lattice)
        n_arena = 0
        if use_cone_runs and arena is not None:
            n_arena, off_lattice = min(off_lattice, arena_cap), 0                 # (nfa_traverse_cone_walk: [1] = arena entries)
        if off_lattice and use_runs and near_hint is not None:
            # rays that are not on the lattice of near_hint (a near plane that differs from it, a march beyond the tabulated
            # sequence: csrc/walk.hip): the same traversal with the per-ray marcher
            return _traverse_samples(rays_o, rays_d, binaries, aabbs, near_planes, far_planes, step_size, cone_angle,
                                     rays_mask=rays_mask, traverse_steps_limit=traverse_steps_limit, t_sorted=t_sorted,
                                     t_indices=t_indices, hits=hits, return_terminate=return_terminate, near_hint=None,
                                     bin_rays=bin_rays, stats_sink=stats_sink, speculate=speculate, n_alive=n_alive,
                                     alive_list=alive_list)
        if stats_sink is not None and c_sum > 0:
            stats_sink["cells_max_over_mean"] = 64.0 * c_max / c_sum
        if SPECULATE and use_runs:
            _SPEC_CAPACITY[spec_key] = ((int(n_sm * 1.03) + 4096) // 4096) * 4096
        if stats_sink is not None and s_sum > 0:
            # how much longer a wave of 64 neighbouring rays runs than its average ray (1 = perfectly coherent)
            stats_sink["max_over_mean"] = 64.0 * s_max / s_sum
        expanded = cap > 0 and n_sm <= cap
        if expanded:
            t_starts, t_ends, ray_indices = t_starts[:n_sm], t_ends[:n_sm], ray_indices[:n_sm]
        else:
            t_starts = torch.empty(n_sm, dtype=torch.float32, device=dev)
            t_ends = torch.empty(n_sm, dtype=torch.float32, device=dev)
            ray_indices = torch.empty(n_sm, dtype=torch.int64, device=dev)
        if n_sm > 0:
            main = torch.cuda.current_stream()
            joined = None
            if (use_runs or use_cone_runs) and n_overflow > 0:
                # The rays whose records did not fit are filled by the serial kernel, a few active lanes per wave for a
                # full-length walk (1.15 ms for 0.1 % of the rays on cfg 5).  It writes ranges the expansion skips, so
                # it runs beside the expansion on a second stream instead of after it.
                sm_starts = packed_info[:, 0].contiguous()
                side = _side_stream(dev)
                side.wait_stream(main)                      # sm_starts, run_cnts and the outputs' allocation
                with torch.cuda.stream(side):
                    a.mode = 1
                    a.terminate_planes = None
                    a.sm_starts = B.ptr(sm_starts)
                    a.sm_t_starts, a.sm_t_ends, a.sm_ray_indices = B.ptr(t_starts), B.ptr(t_ends), B.ptr(ray_indices)
                    a.ray_filter, a.ray_filter_min = B.ptr(run_cnts), MAX_RUNS  # (their counts already honour mask and limit)
                    _launch(a)
                    joined = torch.cuda.Event()
                    joined.record(side)
            if use_runs and not expanded:
                B.call("nfa_expand_runs", n_rays, float(step_size), B.ptr(run_cnts), B.ptr(runs), MAX_RUNS,
                       B.ptr(packed_info), B.ptr(t_starts), B.ptr(t_ends), None, B.ptr(ray_indices), n_sm, B.stream())
            elif use_cone_runs:
                B.call("nfa_expand_cone_runs", n_rays, float(step_size), float(cone_angle), B.ptr(run_cnts), B.ptr(runs),
                       MAX_RUNS, B.ptr(packed_info), B.ptr(t_starts), B.ptr(t_ends), B.ptr(ray_indices), B.stream())
                if n_arena > 0:
                    B.call("nfa_expand_cone_arena", B.ptr(arena), n_arena, float(step_size), float(cone_angle), B.ptr(packed_info),
                           B.ptr(t_starts), B.ptr(t_ends), B.ptr(ray_indices), B.stream())
            if joined is not None:
                main.wait_event(joined)
            if not (use_runs or use_cone_runs):
                a.mode = 1
                a.terminate_planes = None
                sm_starts = packed_info[:, 0].contiguous()
                a.sm_starts = B.ptr(sm_starts)
                a.sm_t_starts, a.sm_t_ends = B.ptr(t_starts), B.ptr(t_ends)
                # the marching kernel writes the distances; the ray ids are a coalesced fill from packed_info
                a.sm_ray_indices = None
                _launch(a)
                B.call("nfa_fill_ray_indices", n_rays, B.ptr(packed_info), B.ptr(ray_indices), B.stream())
    info = tag_trusted(packed_info, n_sm)
    tag_ray_indices(ray_indices, n_rays, info)
    out = (ray_indices, t_starts, t_ends, packed_info)
    return (*out, terminate) if return_terminate else out


def _enlarge_aabb(aabb: Tensor, factor: float) -> Tensor:
    """Scale an aabb about its centre (ref: grid.py:195-198)."""
    center = (aabb[:3] + aabb[3:]) / 2
    extent = (aabb[3:] - aabb[:3]) / 2
    return torch.cat([center - extent * factor, center + extent * factor])


def _query(x: Tensor, data: Tensor, base_aabb: Tensor) -> Tuple[Tensor, Tensor]:
    """Look up multi-level grid values at points ``x`` assuming levels are 2x nested around
    ``base_aabb`` (same contract as the reference's test helper, grid.py:201-237): returns
    (values * inside_selector, inside_selector)."""
    lo, hi = base_aabb[:3], base_aabb[3:]
    u = (x - lo) / (hi - lo)                                # base box -> [0, 1]^3
    r = (u - 0.5).abs().amax(dim=-1).clamp_min(0.1)         # Chebyshev radius; avoid frexp(~0)
    mip = (torch.frexp(r)[1].long() + 1).clamp_min(0)       # 0 inside the base box, +1 per doubling
    inside = mip < data.shape[0]
    uu = (u - 0.5) / (2 ** mip)[:, None] + 0.5
    res = torch.tensor(data.shape[1:], device=x.device)
    ix = torch.minimum((uu * res).long(), res - 1)
    lvl = mip.clamp_max(data.shape[0] - 1)
    return data[lvl, ix[:, 0], ix[:, 1], ix[:, 2]] * inside, inside
