"""Test-mode (inference) marching with ray-level early termination.

Counterpart of the reference's caller harness ``render_image_with_occgrid_test``
(ref: examples/utils.py:252-425).  Same iteration schedule, same per-iteration
semantics and the same final blend:

    n_samples = max(min(num_rays // n_alive, 64), min_samples)          (:338)
    traverse the alive rays for at most n_samples steps, resuming at the previous
        termination planes                                                   (:342-360, :407)
    weights with prefix_trans = 1 - opacity[ray_indices]                    (:370-377)
    in-place accumulation of rgb / opacity / depth                          (:388-405)
    alive = (opacity <= 1 - early_stop_eps) & (samples taken == n_samples)   (:409-414)

The reference implements one iteration as an over-allocated ``traverse_grids`` (dead rays still
occupy a thread and n_alive * n_samples * 3 zero-filled slots), three boolean-index compactions
(three device syncs) and a ``pack_info``.  Here the ray mask and the step limit are honoured inside
the run-length traversal, which emits the compact ``(ray_indices, t_starts, t_ends, packed_info)``
directly: two host reads per iteration (number of alive rays, number of samples); with a constant step ONE -- the schedule
is computed on the device and the host only waits for the sample count it needs to shape the arrays
(PaddedTestModeLoop.render_exact) -- or none (padded=True).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
from torch import Tensor

import ctypes as C

from . import _backend as B
from . import grid as G
from .estimators.occ_grid import OccGridEstimator
from .grid import _traverse_samples, ray_aabb_intersect
from .volrend import accumulate_along_rays_, render_weight_from_density


_VISIBLE_SLOTS = 1024  # include/nerfacc_hip.h: NFA_VISIBLE_SLOTS
ONE_READ = True        # constant step: the exact-shape loop with the schedule on the device (PaddedTestModeLoop.render_exact); False: rounds 1-3's loop


def _render_step_native(seg, t_starts, t_ends, sigmas, rgbs, alpha_thre, rgb, opacity, depth, n_visible) -> None:
    dev = B.require_device(t_starts, t_ends, sigmas, rgbs, rgb, opacity, depth)
    assert rgb.is_contiguous() and opacity.is_contiguous() and depth.is_contiguous()
    with torch.cuda.device(dev):
        B.call("nfa_render_step_accumulate", B.ptr(t_starts), B.ptr(t_ends), B.ptr(sigmas), B.ptr(rgbs),
               B.ptr(seg.packed_info), B.ptr(seg.tiles), seg.n_tiles, seg.n_rays, sigmas.numel(), float(alpha_thre),
               B.ptr(rgb), B.ptr(opacity), B.ptr(depth), B.ptr(n_visible), B.stream())


@torch.no_grad()
def render_rays_test_mode(
    max_samples: int,
    rgb_sigma_fn: Callable,
    estimator: OccGridEstimator,
    rays_o: Tensor,  # [n_rays, 3]
    rays_d: Tensor,  # [n_rays, 3]
    near_plane: float = 0.0,
    far_plane: float = 1e10,
    render_step_size: float = 1e-3,
    render_bkgd: Optional[Tensor] = None,
    cone_angle: float = 0.0,
    alpha_thre: float = 0.0,
    early_stop_eps: float = 1e-4,
    padded: bool = False,
) -> Tuple[Tensor, Tensor, Tensor, int]:
    """Render rays by iterative marching; returns ``(rgb (n,3), opacity (n,1), depth (n,1), total_samples)``.

    ``rgb_sigma_fn(t_starts, t_ends, ray_indices) -> (rgbs (N,3), sigmas (N,))`` as in
    :func:`nerfacc_amd.rendering`.

    ``padded`` (extension, constant step only): the loop without the host in it -- see :class:`PaddedTestModeLoop`.  The
    callback then receives arrays of a FIXED length (the iteration's capacity) whose tail beyond the iteration's samples
    holds earlier, finite values; it must be elementwise in the samples and must not synchronise with the host.
    """
    if cone_angle == 0.0 and rays_o.is_cuda and rays_o.shape[0] > 0 and G._walk_supported(estimator.binaries) and (padded or ONE_READ):
        loop = PaddedTestModeLoop(max_samples, rgb_sigma_fn, estimator, rays_o, rays_d, near_plane, far_plane, render_step_size,
                                  alpha_thre, early_stop_eps)
        return loop.render(render_bkgd) if padded else loop.render_exact(render_bkgd)
    num_rays = rays_o.shape[0]
    device = rays_o.device
    opacity = torch.zeros(num_rays, 1, device=device)
    depth = torch.zeros(num_rays, 1, device=device)
    rgb = torch.zeros(num_rays, 3, device=device)
    ray_mask = torch.ones(num_rays, device=device, dtype=torch.bool)
    min_samples = 1 if cone_angle == 0 else 4  # 1 for synthetic scenes, 4 for real scenes (:312)
    iter_samples = total_samples = 0
    near_planes = torch.full_like(rays_o[..., 0], fill_value=near_plane)
    far_planes = torch.full_like(rays_o[..., 0], fill_value=far_plane)

    n_grids = estimator.binaries.size(0)
    t_sorted = t_indices = hits = None
    if n_grids > 1:  # intersections are computed once (:317-327); one grid is intersected in-kernel
        t_sorted, t_indices, hits = G.ray_events(rays_o, rays_d, estimator.aabbs)
    opc_thre = 1 - early_stop_eps
    n_visible = None  # device counter of the samples that pass alpha_thre on the fused path

    alive = alive_count = None    # the alive rays' ids and their number, written by nfa_alive_rays at the end of an iteration
    n_alive = num_rays
    while iter_samples < max_samples:
        if alive_count is not None:
            n_alive = int(alive_count.item())
        if n_alive == 0:
            break
        n_samples = max(min(num_rays // n_alive, 64), min_samples)
        iter_samples += n_samples

        ray_indices, t_starts, t_ends, packed_info, termination_planes = _traverse_samples(
            rays_o, rays_d, estimator.binaries, estimator.aabbs, near_planes, far_planes, render_step_size,
            cone_angle, rays_mask=ray_mask, traverse_steps_limit=n_samples, t_sorted=t_sorted, t_indices=t_indices,
            hits=hits, return_terminate=True, near_hint=near_plane if iter_samples == n_samples else None, n_alive=n_alive,
            alive_list=alive, speculate=False)   # (an iteration is host-bound: the plain size read is cheaper than the side-stream one)

        n_counted = 0
        if ray_indices.numel() > 0:
            rgbs, sigmas = rgb_sigma_fn(t_starts, t_ends, ray_indices)
            seg = getattr(packed_info, "_nfa_seg", None)
            if (seg is not None and seg[2].contiguous and rgbs.dtype == sigmas.dtype == torch.float32
                    and rgbs.dim() == 2 and rgbs.shape[-1] == 3):
                # weights with prefix_trans = 1 - opacity[ray], alpha_thre masking and the three in-place
                # accumulations (:370-405) as one pass of the segmented engine
                if alpha_thre > 0 and n_visible is None:
                    n_visible = torch.zeros(_VISIBLE_SLOTS, dtype=torch.int64, device=device)
                counter = n_visible if alpha_thre > 0 else None
                # (Until round 3 the pass ran on the rows that have samples once most rays were finished -- gathered before,
                #  scattered back after.  Since a tile owns at most 256 rows the engine walks runs of finished rays for nothing,
                #  and the gathers, scatters and the `nonzero` read-back cost more than they saved: 114 -> 105 ms per cfg-5 image,
                #  17.2 -> 12.5 ms per cfg-2 image without them.)
                _render_step_native(seg[2], t_starts, t_ends, sigmas.contiguous(), rgbs.contiguous(), alpha_thre,
                                    rgb, opacity, depth, counter)
                n_counted = 0 if alpha_thre > 0 else ray_indices.shape[0]
            else:
                weights, _, alphas = render_weight_from_density(
                    t_starts, t_ends, sigmas, packed_info=packed_info, n_rays=num_rays,
                    prefix_trans=1 - opacity[ray_indices].squeeze(-1))
                if alpha_thre > 0:
                    vis_mask = alphas >= alpha_thre
                    ri_v, rgbs, weights, ts_v, te_v = (ray_indices[vis_mask], rgbs[vis_mask], weights[vis_mask],
                                                       t_starts[vis_mask], t_ends[vis_mask])
                else:
                    ri_v, ts_v, te_v = ray_indices, t_starts, t_ends
                accumulate_along_rays_(weights, values=rgbs, ray_indices=ri_v, outputs=rgb)
                accumulate_along_rays_(weights, values=None, ray_indices=ri_v, outputs=opacity)
                accumulate_along_rays_(weights, values=(ts_v + te_v)[..., None] / 2.0, ray_indices=ri_v, outputs=depth)
                n_counted = ri_v.shape[0]
        near_planes = termination_planes
        if packed_info.is_contiguous() and packed_info.dtype == torch.int64 and opacity.dtype == torch.float32:
            # alive = not opaque yet and the whole budget used (:409-414): mask, list and count in one launch
            if alive is None:
                alive = torch.empty(num_rays, dtype=torch.int32, device=device)
                alive_count = torch.zeros(1, dtype=torch.int64, device=device)
            ray_mask = torch.empty(num_rays, dtype=torch.bool, device=device)
            with torch.cuda.device(device):
                B.call("nfa_alive_rays", B.ptr(opacity), B.ptr(packed_info), int(n_samples), float(opc_thre), num_rays,
                       B.ptr(ray_mask), B.ptr(alive), B.ptr(alive_count), B.stream())
        else:
            ray_mask = torch.logical_and(opacity.view(-1) <= opc_thre, packed_info[:, 1] == n_samples)
            alive, alive_count, n_alive = None, None, int(ray_mask.sum().item())
        total_samples += n_counted  # samples that entered the accumulation (:416)

    if n_visible is not None:
        total_samples += int(n_visible.sum().item())
    if render_bkgd is not None:
        rgb = rgb + render_bkgd * (1.0 - opacity)
    eps = torch.finfo(rgb.dtype).eps
    depth = depth / opacity.clamp_min(eps)
    return rgb, opacity, depth, total_samples


class PaddedTestModeLoop:
    """The test-mode loop (ref examples/utils.py:252-425) with the host out of it.

    The exact-shape loop reads two numbers per iteration (alive rays, samples) because the reference's schedule and its
    tensors' shapes depend on them; with ~40 iterations of ~0.1 ms of kernels each, an image of the synthetic scenes costs what
    the HOST needs for 80 synchronisations and ~600 launches (9.7 ms on the build box, 29 ms on a slower host for 4 ms of
    kernels).  Here an iteration has fixed shapes and is driven from the device:

    * ``n_alive * n_samples <= max(n_rays, n_alive * min_samples)`` bounds an iteration's samples, so the sample arrays
      have ONE capacity and the traversal's expansion, the density / colour callback and the accumulation pass run over it
      (entries beyond the iteration's samples are never read by the accumulation: the tile table ends at the real total);
    * the schedule ``n_samples = max(min(n_rays // n_alive, 64), min_samples)`` is computed on the device from the alive
      count (``nfa_testmode_begin``), the walk reads its step limit and the length of the alive list from there
      (``steps_limit_dev``, ``n_listed_dev``), the fill pass for rays with too many run records runs only if there are any
      (``run_if_nonzero``), ``nfa_testmode_alive`` ends the iteration;
    * one iteration is captured into a hipGraph and replayed; the host looks at the alive count every ``check_every``
      replays through a pinned copy, WITHOUT waiting for it: by the time it has queued the next replays the earlier copy has
      landed.  Iterations queued after the last ray died find ``state[0] == 0`` and do nothing.

    Same results as :func:`render_rays_test_mode` (same kernels on the same values; checked in tests/test_gpu_parity.py)."""

    def __init__(self, max_samples, rgb_sigma_fn, estimator, rays_o, rays_d, near_plane=0.0, far_plane=1e10, render_step_size=1e-3,
                 alpha_thre=0.0, early_stop_eps=1e-4, check_every: int = 4, use_graph: bool = True):
        dev = B.require_device(rays_o, rays_d, estimator.binaries)
        self.dev, self.fn, self.max_samples = dev, rgb_sigma_fn, int(max_samples)
        self.alpha_thre, self.opc_thre = float(alpha_thre), 1.0 - float(early_stop_eps)
        self.step, self.near, self.far = float(render_step_size), float(near_plane), float(far_plane)
        self.check_every, self.use_graph = max(1, int(check_every)), bool(use_graph)
        self.rays_o, self.rays_d = rays_o.float().contiguous(), rays_d.float().contiguous()
        self.binaries = (estimator.binaries if estimator.binaries.dtype == torch.bool else estimator.binaries.bool()).contiguous()
        self.aabbs = estimator.aabbs.float().contiguous()
        R = self.R = rays_o.shape[0]
        self.min_samples = 1                                         # cone_angle == 0: synthetic scenes (:312)
        cap = self.cap = max(R * self.min_samples, 4)
        i64, i32, f32 = (dict(dtype=t, device=dev) for t in (torch.int64, torch.int32, torch.float32))
        with torch.cuda.device(dev):
            self.t_sorted = self.t_indices = self.hits = None
            if self.binaries.size(0) > 1:
                self.t_sorted, self.t_indices, hits = G.ray_events(self.rays_o, self.rays_d, self.aabbs)
                self.hits = hits.contiguous()
            self.bits = G._get_walk_bits(self.binaries)
            self.planes = torch.empty(R, **f32)                      # near planes in, termination planes out (per ray, in place)
            self.far_planes = torch.full((R,), self.far, **f32)
            self.sm_cnts, self.run_cnts = torch.empty(R, **i64), torch.empty(R, **i32)
            self.runs = torch.empty((G.MAX_RUNS, R), **i64)
            self.packed_info = torch.empty((R, 2), **i64)
            self.scratch = B.cumsum_scratch(R, dev)
            self.meta = torch.zeros(8, **i64)                        # [total, coherence sums (2), rays with too many runs, -]
            self.t_starts, self.t_ends = torch.zeros(cap, **f32), torch.zeros(cap, **f32)
            self.ray_indices = torch.zeros(cap, **i64)
            self.tile_elems, self.n_tiles = B.seg_plan(cap, R)
            self.tiles = torch.empty((int(B.load().nfa_seg_table_rows(self.n_tiles)), 2), **i64)
            self.ray_mask = torch.empty(R, dtype=torch.bool, device=dev)
            self.alive = torch.empty(R, **i32)
            self.alive_count = torch.empty(2, **i64)        # [alive now, alive at the start of the iteration]
            self.state = torch.empty(8, **i32)
            self.n_visible = torch.zeros(_VISIBLE_SLOTS, **i64) if self.alpha_thre > 0 else None
            self.rgb, self.opacity, self.depth = torch.empty(R, 3, **f32), torch.empty(R, 1, **f32), torch.empty(R, 1, **f32)
            self.host = torch.empty(2, dtype=torch.int64, pin_memory=True)
            a = G._traverse_args(self.rays_o, self.rays_d, self.ray_mask, self.binaries, self.aabbs, self.t_sorted, self.t_indices,
                                 self.hits, self.planes, self.far_planes, self.step, 0.0, 64, 2)
            a.sm_cnts, a.terminate_planes = B.ptr(self.sm_cnts), B.ptr(self.planes)
            a.steps_limit_dev, a.n_listed_dev = B.ptr(self.state), B.ptr(self.alive_count[1:2])
            self.args = a
            f = G._traverse_args(self.rays_o, self.rays_d, self.ray_mask, self.binaries, self.aabbs, self.t_sorted, self.t_indices,
                                 self.hits, self.planes, self.far_planes, self.step, 0.0, 64, 1)
            # NOTE: the fill pass re-walks a ray from its near plane; it reads the planes BEFORE the walk overwrote them, so
            # it gets its own copy of this iteration's near planes
            self.planes_in = torch.empty(R, **f32)
            f.near_planes = B.ptr(self.planes_in)
            f.sm_cnts = B.ptr(self.sm_cnts)
            f.sm_t_starts, f.sm_t_ends, f.sm_ray_indices = B.ptr(self.t_starts), B.ptr(self.t_ends), B.ptr(self.ray_indices)
            f.ray_filter, f.ray_filter_min = B.ptr(self.run_cnts), G.MAX_RUNS
            f.steps_limit_dev, f.run_if_nonzero = B.ptr(self.state), B.ptr(self.meta[3:4])
            self.fill_args = f
            self.sm_starts = torch.empty(R, **i64)
            f.sm_starts = B.ptr(self.sm_starts)
            self.graph = None

    def _reset(self) -> None:
        self.planes.fill_(self.near)
        self.ray_mask.fill_(True)
        torch.arange(self.R, out=self.alive)
        self.alive_count.fill_(self.R)
        self.state.zero_()
        self.rgb.zero_(); self.opacity.zero_(); self.depth.zero_()
        if self.n_visible is not None:
            self.n_visible.zero_()

    def _iteration(self) -> None:
        """One iteration, no host reads, fixed shapes (capturable)."""
        R, cap, s = self.R, self.cap, B.stream
        B.call("nfa_testmode_begin", B.ptr(self.alive_count), B.ptr(self.state), R, self.min_samples, self.max_samples,
               B.ptr(self.sm_cnts), B.ptr(self.run_cnts), B.ptr(self.meta), self.meta.numel(), s())
        self.planes_in.copy_(self.planes)
        B.call("nfa_traverse_runs", C.byref(self.args), B.ptr(self.bits), B.ptr(self.run_cnts), B.ptr(self.runs), G.MAX_RUNS,
               B.ptr(self.meta[3:4]), float("nan"), B.ptr(self.alive), R, s())
        B.call("nfa_exclusive_cumsum_pairs_stats_i64", B.ptr(self.sm_cnts), R, B.ptr(self.packed_info), B.ptr(self.meta[0:3]),
               B.ptr(self.scratch), s())
        B.call("nfa_expand_runs", R, self.step, B.ptr(self.run_cnts), B.ptr(self.runs), G.MAX_RUNS, B.ptr(self.packed_info),
               B.ptr(self.t_starts), B.ptr(self.t_ends), None, B.ptr(self.ray_indices), cap, s())
        self.sm_starts.copy_(self.packed_info[:, 0])
        B.call("nfa_traverse_grids", C.byref(self.fill_args), s())          # rays with more than MAX_RUNS runs, if any
        B.call("nfa_seg_build_tiles", B.ptr(self.packed_info), R, cap, self.tile_elems, self.n_tiles, B.ptr(self.tiles), None, s())
        rgbs, sigmas = self.fn(self.t_starts, self.t_ends, self.ray_indices)
        B.call("nfa_render_step_accumulate", B.ptr(self.t_starts), B.ptr(self.t_ends), B.ptr(sigmas.contiguous()),
               B.ptr(rgbs.contiguous()), B.ptr(self.packed_info), B.ptr(self.tiles), self.n_tiles, R, cap, self.alpha_thre,
               B.ptr(self.rgb), B.ptr(self.opacity), B.ptr(self.depth), B.ptr(self.n_visible), s())
        B.call("nfa_testmode_alive", B.ptr(self.opacity), B.ptr(self.packed_info), B.ptr(self.state), self.opc_thre, R,
               B.ptr(self.ray_mask), B.ptr(self.alive), B.ptr(self.alive_count), 0 if self.alpha_thre > 0 else 1, s())

    @torch.no_grad()
    def render_exact(self, render_bkgd: Optional[Tensor] = None):
        """The same loop with EXACT shapes for the callback (the reference's contract: ``rgb_sigma_fn`` sees the iteration's
        samples and nothing else) and ONE host read per iteration instead of two: the schedule still lives on the device
        (the walk reads its step limit and the length of the alive list there), so the host only waits for the iteration's
        sample count -- which it needs to shape the arrays -- and learns the step limit with it.  The exact-shape loop of
        rounds 1-3 waited for the alive count before the walk and for the sample count behind it; on a slow host that was
        29 ms per 1 M-ray image for 4 ms of kernels."""
        R, s = self.R, B.stream
        i64, f32 = dict(dtype=torch.int64, device=self.dev), dict(dtype=torch.float32, device=self.dev)
        with torch.cuda.device(self.dev):
            self._reset()
            if getattr(self, "host_n", None) is None:
                self.host_n = torch.empty(1, dtype=torch.int64, pin_memory=True)
                self.host_s = torch.empty(1, dtype=torch.int32, pin_memory=True)
            budget = -(-self.max_samples // self.min_samples)
            f = self.fill_args
            self.iterations_run = 0
            while budget > 0:
                budget -= 1
                B.call("nfa_testmode_begin", B.ptr(self.alive_count), B.ptr(self.state), R, self.min_samples, self.max_samples,
                       B.ptr(self.sm_cnts), B.ptr(self.run_cnts), B.ptr(self.meta), self.meta.numel(), s())
                self.planes_in.copy_(self.planes)
                B.call("nfa_traverse_runs", C.byref(self.args), B.ptr(self.bits), B.ptr(self.run_cnts), B.ptr(self.runs), G.MAX_RUNS,
                       B.ptr(self.meta[3:4]), float("nan"), B.ptr(self.alive), R, s())
                B.call("nfa_exclusive_cumsum_pairs_stats_i64", B.ptr(self.sm_cnts), R, B.ptr(self.packed_info),
                       B.ptr(self.meta[0:3]), B.ptr(self.scratch), s())
                # the one read: the iteration's samples and its step limit (0: the loop is over)
                self.host_n.copy_(self.meta[0:1], non_blocking=True)
                self.host_s.copy_(self.state[0:1], non_blocking=True)
                ev = torch.cuda.Event(); ev.record(); ev.synchronize()
                n, n_samples = int(self.host_n[0]), int(self.host_s[0])
                if n_samples == 0:
                    break
                self.iterations_run += 1
                if n > 0:
                    t_starts, t_ends = torch.empty(n, **f32), torch.empty(n, **f32)
                    ray_indices = torch.empty(n, **i64)
                    B.call("nfa_expand_runs", R, self.step, B.ptr(self.run_cnts), B.ptr(self.runs), G.MAX_RUNS,
                           B.ptr(self.packed_info), B.ptr(t_starts), B.ptr(t_ends), None, B.ptr(ray_indices), n, s())
                    self.sm_starts.copy_(self.packed_info[:, 0])
                    f.sm_t_starts, f.sm_t_ends, f.sm_ray_indices = B.ptr(t_starts), B.ptr(t_ends), B.ptr(ray_indices)
                    B.call("nfa_traverse_grids", C.byref(f), s())          # rays with more than MAX_RUNS runs, if any
                    tile_elems, n_tiles = B.seg_plan(n, R)
                    B.call("nfa_seg_build_tiles", B.ptr(self.packed_info), R, n, tile_elems, n_tiles, B.ptr(self.tiles), None, s())
                    rgbs, sigmas = self.fn(t_starts, t_ends, ray_indices)
                    rgbs, sigmas = rgbs.float().contiguous(), sigmas.float().contiguous()
                    B.call("nfa_render_step_accumulate", B.ptr(t_starts), B.ptr(t_ends), B.ptr(sigmas), B.ptr(rgbs),
                           B.ptr(self.packed_info), B.ptr(self.tiles), n_tiles, R, n, self.alpha_thre, B.ptr(self.rgb),
                           B.ptr(self.opacity), B.ptr(self.depth), B.ptr(self.n_visible), s())
                B.call("nfa_testmode_alive", B.ptr(self.opacity), B.ptr(self.packed_info), B.ptr(self.state), self.opc_thre, R,
                       B.ptr(self.ray_mask), B.ptr(self.alive), B.ptr(self.alive_count), 0 if self.alpha_thre > 0 else 1, s())
            f.sm_t_starts, f.sm_t_ends, f.sm_ray_indices = B.ptr(self.t_starts), B.ptr(self.t_ends), B.ptr(self.ray_indices)
            st = self.state.cpu()
            total = int(st[4:6].view(torch.int64)[0]) if self.n_visible is None else int(self.n_visible.sum().item())
            rgb, opacity, depth = self.rgb.clone(), self.opacity.clone(), self.depth.clone()
            if render_bkgd is not None:
                rgb = rgb + render_bkgd * (1.0 - opacity)
            depth = depth / opacity.clamp_min(torch.finfo(rgb.dtype).eps)
            return rgb, opacity, depth, total

    @torch.no_grad()
    def render(self, render_bkgd: Optional[Tensor] = None):
        with torch.cuda.device(self.dev):
            if self.use_graph and self.graph is None:
                # warm-up on a side stream (allocator pools, lazily built library state), then capture ONE iteration
                self._reset()
                side = torch.cuda.Stream(device=self.dev)
                side.wait_stream(torch.cuda.current_stream(self.dev))
                with torch.cuda.stream(side):
                    self._iteration()
                torch.cuda.current_stream(self.dev).wait_stream(side)
                torch.cuda.synchronize(self.dev)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self._iteration()
            self._reset()
            # at most max_samples / min_samples iterations (every one hands out at least min_samples samples per ray)
            budget = -(-self.max_samples // self.min_samples)
            pending = []                                     # (event, pinned slot) of the alive counts on their way to the host
            done = False
            self.iterations_queued = 0
            while not done and budget > 0:
                for _ in range(min(self.check_every, budget)):
                    self.graph.replay() if self.graph is not None else self._iteration()
                    budget -= 1
                    self.iterations_queued += 1
                slot = len(pending) & 1
                self.host[slot:slot + 1].copy_(self.alive_count[0:1], non_blocking=True)
                ev = torch.cuda.Event(); ev.record()
                pending.append((ev, slot))
                if len(pending) >= 2:                        # the copy queued a batch ago has landed by now (or nearly)
                    ev0, slot0 = pending.pop(0)
                    ev0.synchronize()
                    done = int(self.host[slot0]) == 0
            torch.cuda.synchronize(self.dev)
            st = self.state.cpu()
            total = int(st[4:6].view(torch.int64)[0]) if self.n_visible is None else int(self.n_visible.sum().item())
            self.iterations_run = int(st[2])
            rgb, opacity, depth = self.rgb.clone(), self.opacity.clone(), self.depth.clone()
            if render_bkgd is not None:
                rgb = rgb + render_bkgd * (1.0 - opacity)
            depth = depth / opacity.clamp_min(torch.finfo(rgb.dtype).eps)
            return rgb, opacity, depth, total
