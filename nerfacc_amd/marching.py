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
directly: two host reads per iteration (number of alive rays, number of samples).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
from torch import Tensor

from . import _backend as B
from .estimators.occ_grid import OccGridEstimator
from .grid import _traverse_samples, ray_aabb_intersect
from .volrend import accumulate_along_rays_, render_weight_from_density


_VISIBLE_SLOTS = 1024  # include/nerfacc_hip.h: NFA_VISIBLE_SLOTS


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
) -> Tuple[Tensor, Tensor, Tensor, int]:
    """Render rays by iterative marching; returns ``(rgb (n,3), opacity (n,1), depth (n,1), total_samples)``.

    ``rgb_sigma_fn(t_starts, t_ends, ray_indices) -> (rgbs (N,3), sigmas (N,))`` as in
    :func:`nerfacc_amd.rendering`.
    """
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
        t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, estimator.aabbs)
        t_sorted, t_indices = torch.sort(torch.cat([t_mins, t_maxs], -1), -1)
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
