"""Occupancy-grid estimator (ref: nerfacc/estimators/occ_grid.py).

``sampling`` is the hot path (ref :85-221): traversal -> user density callback -> visibility ->
compaction.  The reference does this with 5 boolean-index gathers (5 device syncs) and an
``occs.mean().item()``; here the traversal emits ``(ray_indices, t_starts, t_ends)`` directly,
visibility and compaction are two fused passes with one size read-back, and the mean occupancy
is cached per ``occs`` version.  Buffer names / dtypes match the reference so ``state_dict``s
interchange (``resolution``, ``aabbs``, ``occs``, ``binaries``).

Grid maintenance: ``_update`` (ref :368-404) runs as native passes on a ROCm device (csrc/gridupd.hip) and leaves the
traversal's bit-packed grid copy beside the ``torch.bool`` buffer; the cell selection and ``mark_invisible_cells`` are torch.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Tuple, Union

import torch
from torch import Tensor

from .. import _backend as B
from .._segments import SegInfo, tag_ray_indices, tag_trusted
from ..grid import _enlarge_aabb, _exclusive_cumsum, _traverse_samples
from ..volrend import _visibility_native
from .base import AbstractEstimator


class TraversalHandle:
    """Result of :meth:`OccGridEstimator.prefetch_traversal`: (ray_indices, t_starts, t_ends, packed_info) produced on
    a side stream plus the event that marks their completion."""

    def __init__(self, tensors, event, key):
        self._tensors, self._event, self.key = tensors, event, key

    def consume(self, stream):
        assert self._tensors is not None, "a TraversalHandle can be consumed once"
        stream.wait_event(self._event)
        tensors, self._tensors = self._tensors, None
        for t in tensors:  # allocated on the side stream, used on `stream` from now on
            t.record_stream(stream)
        seg = getattr(tensors[3], "_nfa_seg", None)
        if seg is not None and seg[2].tiles is not None:
            seg[2].tiles.record_stream(stream)
        return tensors


class OccGridEstimator(AbstractEstimator):
    """Occupancy grid transmittance estimator for spatial skipping ("Instant-NGP" style).

    Args:
        roi_aabb: region of interest ``[xmin, ymin, zmin, xmax, ymax, zmax]``.
        resolution: cells per axis (int, list or tensor of 3). Default 128.
        levels: number of nested grids, level ``i`` covering ``roi_aabb`` scaled by ``2**i``.
    """

    DIM: int = 3

    def __init__(self, roi_aabb: Union[List[int], Tensor], resolution: Union[int, List[int], Tensor] = 128,
                 levels: int = 1, **kwargs) -> None:
        super().__init__()
        if "contraction_type" in kwargs:
            raise ValueError("`contraction_type` is not supported anymore for nerfacc >= 0.4.0.")
        if isinstance(resolution, int):
            resolution = [resolution] * self.DIM
        if isinstance(resolution, (list, tuple)):
            resolution = torch.tensor(resolution, dtype=torch.int32)
        assert isinstance(resolution, Tensor), f"Invalid type: {resolution}!"
        assert resolution.shape[0] == self.DIM, f"Invalid shape: {resolution}!"
        if isinstance(roi_aabb, (list, tuple)):
            roi_aabb = torch.tensor(roi_aabb, dtype=torch.float32)
        assert isinstance(roi_aabb, Tensor), f"Invalid type: {roi_aabb}!"
        assert roi_aabb.shape[0] == self.DIM * 2, f"Invalid shape: {roi_aabb}!"

        self.levels = levels
        self.cells_per_lvl = int(resolution.prod().item())
        aabbs = torch.stack([_enlarge_aabb(roi_aabb, 2 ** i) for i in range(levels)], dim=0)
        # persistent state (same names as the reference, occ_grid.py:67-75)
        self.register_buffer("resolution", resolution)
        self.register_buffer("aabbs", aabbs)
        self.register_buffer("occs", torch.zeros(levels * self.cells_per_lvl))
        self.register_buffer("binaries", torch.zeros([levels] + resolution.tolist(), dtype=torch.bool))
        # derived helpers
        self.register_buffer("grid_coords", _meshgrid3d(resolution).reshape(self.cells_per_lvl, self.DIM),
                             persistent=False)
        self.register_buffer("grid_indices", torch.arange(self.cells_per_lvl), persistent=False)
        self._occs_mean_cache = None
        self._prefetch_stream = None
        self._walk_stats = {}  # coherence of the previous batches' rays (see bin_rays)

    # ------------------------------------------------------------------ hot path
    def _planes(self, rays_o: Tensor, near_plane: float, far_plane: float):
        """Constant near / far plane tensors (ref :161-162 builds them with full_like on every call); cached per
        (n_rays, values, device) and never written in place."""
        key = (rays_o.shape[0], float(near_plane), float(far_plane), rays_o.device, rays_o.dtype)
        cached = getattr(self, "_planes_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, torch.full_like(rays_o[..., 0], fill_value=near_plane),
                      torch.full_like(rays_o[..., 0], fill_value=far_plane))
            self._planes_cache = cached
        return cached[1], cached[2]

    #: Batches of unrelated rays (random pixels of random images, the usual training batch) are walked ~1.6x faster when
    #: rays of similar path length share a wave (``nfa_bin_rays``); image-ordered rays are coherent already and gain
    #: nothing.  None (default): decided from the sample counts of the previous batch (a wave of 64 neighbouring rays
    #: running more than five times as long as its average ray); True / False force it.  Results are identical either way.
    #: (Extension: the reference has no such switch.)
    bin_rays: Optional[bool] = None

    def _occs_mean(self) -> float:
        """``self.occs.mean().item()`` (ref :183) cached until ``occs`` changes."""
        # the tensor object itself is part of the key: a re-assigned buffer may reuse the address of the old one at version 0
        key = (id(self.occs), self.occs.data_ptr(), self.occs._version)
        if self._occs_mean_cache is None or self._occs_mean_cache[0] != key:
            self._occs_mean_cache = (key, float(self.occs.mean().item()))
        return self._occs_mean_cache[1]

    def _traverse(self, rays_o, rays_d, near_plane, far_plane, t_min, t_max, render_step_size, stratified, cone_angle,
                  speculate=True):
        near_planes, far_planes = self._planes(rays_o, near_plane, far_plane)
        if t_min is not None:
            near_planes = torch.clamp(near_planes, min=t_min)
        if t_max is not None:
            far_planes = torch.clamp(far_planes, max=t_max)
        if stratified:
            near_planes = near_planes + torch.rand_like(near_planes) * render_step_size
        use_bins = self.bin_rays
        if use_bins is None and not cone_angle > 0.0:
            # automatic: decided by the coherence the previous batches showed (sample counts of 64 neighbouring rays; image
            # order: 1.6-2.5, random rays: ~12).  With a cone angle the walk decides itself, from the coherence of its own
            # binning key (cells crossed): None is passed on.
            use_bins = self._walk_stats.get("max_over_mean", 1.0) > 5.0 and rays_o.shape[0] >= 65536
        return _traverse_samples(rays_o, rays_d, self.binaries, self.aabbs, near_planes, far_planes, render_step_size,
                                 cone_angle, near_hint=near_plane, bin_rays=None if use_bins is None else bool(use_bins),
                                 stats_sink=self._walk_stats, speculate=speculate)

    def _traversal_key(self, rays_o, rays_d, near_plane, far_plane, t_min, t_max, render_step_size, stratified, cone_angle):
        """Identity of everything the geometric half of ``sampling`` reads: tensors by (address, shape, version counter)
        -- rays rewritten in place after a prefetch (the usual double-buffer pattern) no longer match."""
        def tid(t):
            return None if t is None else (t.data_ptr(), tuple(t.shape), t._version)
        # (the grid also by object identity: a re-assigned ``binaries`` may reuse the old one's address at version 0)
        return (tid(rays_o), tid(rays_d), tid(t_min), tid(t_max), float(near_plane), float(far_plane),
                float(render_step_size), bool(stratified), float(cone_angle), tid(self.binaries), id(self.binaries))

    @torch.no_grad()
    def prefetch_traversal(
        self,
        rays_o: Tensor,
        rays_d: Tensor,
        near_plane: float = 0.0,
        far_plane: float = 1e10,
        t_min: Optional[Tensor] = None,
        t_max: Optional[Tensor] = None,
        render_step_size: float = 1e-3,
        stratified: bool = False,
        cone_angle: float = 0.0,
        wait_for_inputs: bool = True,
    ) -> "TraversalHandle":
        """Run the geometric half of :meth:`sampling` (grid traversal; no density callback) for a batch of rays on
        a side stream and return a handle for ``sampling(..., traversal=handle)``.

        The walk through the grid is bound by instruction issue, the rendering passes by HBM bandwidth: issued on two
        streams they overlap, so the traversal of the NEXT batch can run under the rendering / backward of the
        current one (pipeline objects that carry sizes across calls).  The result is exactly what
        ``sampling`` would compute itself (same kernels) as long as the occupancy grid is not modified in between.
        ``wait_for_inputs=False`` skips the dependency on the current stream's pending work -- only when the inputs
        are already resident (otherwise the side stream would wait for the very work it is meant to overlap with).
        """
        if self._prefetch_stream is None:
            # a high-priority stream: it comes from a different pool than ordinary streams, which makes it far less
            # likely to share a hardware queue with the stream it is supposed to overlap with
            try:
                self._prefetch_stream = torch.cuda.Stream(device=rays_o.device, priority=-1)
            except Exception:  # pragma: no cover - priorities unsupported
                self._prefetch_stream = torch.cuda.Stream(device=rays_o.device)
        side = self._prefetch_stream
        if wait_for_inputs:
            side.wait_stream(torch.cuda.current_stream(rays_o.device))
        with torch.cuda.stream(side):
            # (on the side stream the size read is off the critical path already: no speculative expansion)
            out = self._traverse(rays_o, rays_d, near_plane, far_plane, t_min, t_max, render_step_size, stratified,
                                 cone_angle, speculate=False)
            event = torch.cuda.Event()
            event.record(side)
        key = self._traversal_key(rays_o, rays_d, near_plane, far_plane, t_min, t_max, render_step_size, stratified, cone_angle)
        return TraversalHandle(out, event, key)

    @torch.no_grad()
    def sampling(
        self,
        rays_o: Tensor,  # [n_rays, 3]
        rays_d: Tensor,  # [n_rays, 3]
        sigma_fn: Optional[Callable] = None,
        alpha_fn: Optional[Callable] = None,
        near_plane: float = 0.0,
        far_plane: float = 1e10,
        t_min: Optional[Tensor] = None,  # [n_rays]
        t_max: Optional[Tensor] = None,  # [n_rays]
        render_step_size: float = 1e-3,
        early_stop_eps: float = 1e-4,
        alpha_thre: float = 0.0,
        stratified: bool = False,
        cone_angle: float = 0.0,
        traversal: Optional["TraversalHandle"] = None,
    ) -> Tuple[Tensor, Tensor, Tensor]:
        """Sampling with spatial skipping; not differentiable.

        Arguments as the reference (occ_grid.py:86-148).  Returns ``(ray_indices LongTensor
        (n_samples,), t_starts (n_samples,), t_ends (n_samples,))``, ray-sorted.  ``sigma_fn`` /
        ``alpha_fn`` take ``(t_starts, t_ends, ray_indices)`` and return densities / opacities
        ``(N,)``; when given (and a threshold is active) invisible samples are dropped.
        """
        if traversal is not None:
            key = self._traversal_key(rays_o, rays_d, near_plane, far_plane, t_min, t_max, render_step_size, stratified, cone_angle)
            if traversal.key != key:
                raise ValueError("nerfacc_amd: the prefetched traversal was made for other rays / planes / step / grid "
                                 "(or the ray / plane tensors were modified in place since)")
            ray_indices, t_starts, t_ends, packed_info = traversal.consume(torch.cuda.current_stream(rays_o.device))
        else:
            ray_indices, t_starts, t_ends, packed_info = self._traverse(
                rays_o, rays_d, near_plane, far_plane, t_min, t_max, render_step_size, stratified, cone_angle)

        if (alpha_thre > 0.0 or early_stop_eps > 0.0) and (sigma_fn is not None or alpha_fn is not None):
            alpha_thre = min(alpha_thre, self._occs_mean())
            seg: SegInfo = tag_trusted(packed_info, t_starts.numel())
            if sigma_fn is not None:
                if t_starts.shape[0] != 0:
                    sigmas = sigma_fn(t_starts, t_ends, ray_indices)
                else:
                    sigmas = torch.empty((0,), device=t_starts.device)
                assert sigmas.shape == t_starts.shape, "sigmas must have shape of (N,)! Got {}".format(sigmas.shape)
                vis, cnts = _visibility_native(seg, t_starts, t_ends, sigmas, None, early_stop_eps, alpha_thre, True)
            else:
                if t_starts.shape[0] != 0:
                    alphas = alpha_fn(t_starts, t_ends, ray_indices)
                else:
                    alphas = torch.empty((0,), device=t_starts.device)
                assert alphas.shape == t_starts.shape, "alphas must have shape of (N,)! Got {}".format(alphas.shape)
                vis, cnts = _visibility_native(seg, None, None, alphas, None, early_stop_eps, alpha_thre, True)
            compacted = _compact(seg, vis, cnts, t_starts, t_ends)
            if compacted is not None:
                ray_indices, t_starts, t_ends = compacted
        return ray_indices, t_starts, t_ends

    # ------------------------------------------------------------------ grid maintenance (torch)
    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2,
                             ema_decay: float = 0.95, warmup_steps: int = 256, n: int = 16) -> None:
        """EMA-update the grid every ``n`` training steps (ref :223-259)."""
        if not self.training:
            raise RuntimeError(
                "You should only call this function only during training. "
                "Please call _update() directly if you want to update the field during inference.")
        if step % n == 0:
            self._update(step=step, occ_eval_fn=occ_eval_fn, occ_thre=occ_thre, ema_decay=ema_decay,
                         warmup_steps=warmup_steps)

    @torch.no_grad()
    def mark_invisible_cells(self, K: Tensor, c2w: Tensor, width: int, height: int, near_plane: float = 0.0,
                             chunk: int = 32 ** 3) -> None:
        """Set ``occs = -1`` for cells no camera sees (or that sit in front of a camera's near
        plane); run once before training (ref :262-332)."""
        assert K.dim() == 3 and K.shape[1:] == (3, 3)
        assert c2w.dim() == 3 and (c2w.shape[1:] == (3, 4) or c2w.shape[1:] == (4, 4))
        assert K.shape[0] == c2w.shape[0] or K.shape[0] == 1
        n_cams = c2w.shape[0]
        rot = c2w[:, :3, :3].transpose(2, 1)       # world -> camera rotation
        trans = -rot @ c2w[:, :3, 3:]              # world -> camera translation
        for lvl, indices in enumerate(self._get_all_cells()):
            coords = self.grid_coords[indices]
            lo, hi = self.aabbs[lvl, :3], self.aabbs[lvl, 3:]
            for i in range(0, len(indices), chunk):
                ids = indices[i:i + chunk]
                world = (lo + coords[i:i + chunk] / (self.resolution - 1) * (hi - lo)).T  # (3, chunk)
                uvd = K @ (rot @ world + trans)                                              # (n_cams, 3, chunk)
                depth = uvd[:, 2]
                uv = uvd[:, :2] / uvd[:, 2:]
                in_image = (depth >= 0) & (uv[:, 0] >= 0) & (uv[:, 0] < width) & (uv[:, 1] >= 0) & (uv[:, 1] < height)
                seen = ((depth >= near_plane) & in_image).sum(0) / n_cams > 0
                too_near = ((depth < near_plane) & in_image).any(0)
                self.occs[lvl * self.cells_per_lvl + ids] = torch.where(seen & ~too_near, 0.0, -1.0)

    @torch.no_grad()
    def _get_all_cells(self) -> List[Tensor]:
        """Per level, the indices of cells not marked invisible (occs >= 0)."""
        out = []
        for lvl in range(self.levels):
            keep = self.occs[lvl * self.cells_per_lvl + self.grid_indices] >= 0.0
            out.append(self.grid_indices[keep])
        return out

    @torch.no_grad()
    def _sample_uniform_and_occupied_cells(self, n: int) -> List[Tensor]:
        """Per level: n uniformly drawn (visible) cells plus the occupied cells -- all of them, or n drawn with replacement
        when there are more than n (ref :345-366).

        The reference filters with a boolean index and a ``nonzero`` per level (each a device synchronisation, three per
        level).  Here the two selections are stable compactions by prefix sum + scatter for all levels at once, with ONE
        host read (the levels' output sizes); same cells in the same order as the reference's expressions."""
        if self.occs.is_cuda and self.levels * self.cells_per_lvl <= self.ONE_READ_MAX_CELLS:
            return self._sample_cells_one_read(n)
        # (CPU tensors, and grids so large that the one-read form's all-level prefix sums would take gigabytes -- 4 x 512^3:
        #  the reference's per-level expressions, whose temporaries are one level's draws and its occupied cells)
        dev, L, cells = self.occs.device, self.levels, self.cells_per_lvl
        out = []
        for lvl in range(L):
            uni = torch.randint(cells, (n,), device=dev)
            uni = uni[self.occs[lvl * cells + uni] >= 0.0]
            occ = torch.nonzero(self.binaries[lvl].flatten())[:, 0]
            if n < len(occ):
                occ = occ[torch.randint(len(occ), (n,), device=dev)]
            out.append(torch.cat([uni, occ], dim=0))
        return out

    #: largest grid (cells over all levels) whose cell selection runs in the one-read form: its prefix sums and draws are
    #: int32 arrays over all levels at once (2^27 cells: ~1.3 GB of temporaries)
    ONE_READ_MAX_CELLS = 1 << 27

    @torch.no_grad()
    def _sample_cells_one_read(self, n: int) -> List[Tensor]:
        dev, L, cells = self.occs.device, self.levels, self.cells_per_lvl
        uni = torch.randint(cells, (L, n), device=dev)                                     # ref :350
        uni_ok = self.occs.view(L, cells).gather(1, uni) >= 0.0                             # ref :352-353
        # (int32: a level has fewer than 2^31 cells; half the memory of the default int64 prefix sums)
        uni_pos = torch.cumsum(uni_ok, dim=1, dtype=torch.int32)                            # 1-based slot of every kept draw
        occ_flag = self.binaries.view(L, cells)
        occ_pos = torch.cumsum(occ_flag, dim=1, dtype=torch.int32)
        sizes = torch.stack([uni_pos[:, -1], occ_pos[:, -1]], dim=1).tolist()              # the one device -> host read
        out = []
        cell_ids = torch.arange(cells, device=dev)
        for lvl, (n_uni, n_occ) in enumerate(sizes):
            n_uni, n_occ = int(n_uni), int(n_occ)
            if n < n_occ:   # ref :356-360: n of the occupied cells, drawn with replacement = the k-th occupied cell for random k
                k = torch.randint(n_occ, (n,), device=dev)
                occ = torch.searchsorted(occ_pos[lvl], (k + 1).to(torch.int32))             # first cell whose count reaches k + 1
                n_take = n
            else:
                occ, n_take = None, n_occ
            res = torch.empty(n_uni + n_take + 1, dtype=torch.int64, device=dev)            # (+1: the slot dropped entries land in)
            dump = n_uni + n_take
            res.scatter_(0, torch.where(uni_ok[lvl], uni_pos[lvl] - 1, dump).long(), uni[lvl])
            if occ is None:
                res.scatter_(0, torch.where(occ_flag[lvl], n_uni + occ_pos[lvl] - 1, dump).long(), cell_ids)
            else:
                res[n_uni:dump] = occ
            out.append(res[:dump])
        return out

    #: Process group over which :meth:`_update` keeps the grid identical on all ranks (``None``: no communication, the
    #: reference's behaviour -- it is single-device).  OPT-IN, because it makes every call of ``update_every_n_steps`` /
    #: ``_update`` a COLLECTIVE: all ranks of the group must call it at the same steps (same ``n``, same ``training`` flag),
    #: with estimators that describe the same scene.  Extension: the reference has no such attribute.
    sync_group = None

    @torch.no_grad()
    def _update(self, step: int, occ_eval_fn: Callable, occ_thre: float = 0.01, ema_decay: float = 0.95,
                warmup_steps: int = 256) -> None:
        """occs = max(occs * decay, occ(x)) at jittered cell positions, then re-binarise (ref :368-404).

        On a ROCm device the three stages are native passes (csrc/gridupd.hip): cell ids + jitter -> positions, the EMA /
        max scatter, and a device-side threshold + binarisation that writes the ``torch.bool`` buffer (the serialised
        view) together with the traversal's bit-packed copy.

        With :attr:`sync_group` set, the ranks of that group end every update with the same ``occs`` / ``binaries`` whatever
        cells each of them sampled: the per-cell maximum of the occupancies the ranks evaluated is MAX-all-reduced
        and the decay is applied to the UNION of the sampled cells -- ``occs = max(occs * decay, max over ranks of occ)``
        wherever some rank sampled the cell -- so the grid is pruned as fast as on one device (a MAX over the ranks' already
        updated ``occs`` would decay a cell only when every rank happened to sample it).  One all-reduce of ``occs``' size."""
        if step < warmup_steps:
            lvl_indices = self._get_all_cells()
        else:
            lvl_indices = self._sample_uniform_and_occupied_cells(self.cells_per_lvl // 4)
        group = self.sync_group
        if group is not None and torch.distributed.get_world_size(group) > 1:
            cand = torch.full_like(self.occs, float("-inf"))     # per cell: the largest occupancy this rank evaluated
            for lvl, indices in enumerate(lvl_indices):
                jitter = torch.rand((indices.shape[0], self.DIM), dtype=torch.float32, device=indices.device)   # ref :385
                occ = occ_eval_fn(self._cell_points(lvl, indices, jitter)).squeeze(-1).to(torch.float32)
                cand.scatter_reduce_(0, lvl * self.cells_per_lvl + indices, occ, reduce="amax", include_self=True)
            torch.distributed.all_reduce(cand, op=torch.distributed.ReduceOp.MAX, group=group)
            sampled = cand > float("-inf")
            self.occs = torch.where(sampled, torch.maximum(self.occs * ema_decay, cand), self.occs)
            self._occs_mean_cache = None
        else:
            for lvl, indices in enumerate(lvl_indices):
                jitter = torch.rand((indices.shape[0], self.DIM), dtype=torch.float32, device=indices.device)   # ref :385
                x = self._cell_points(lvl, indices, jitter)
                occ = occ_eval_fn(x).squeeze(-1)
                self._ema_update(lvl, indices, occ, ema_decay)
        self._rebinarize(occ_thre)

    def _cell_points(self, lvl: int, indices: Tensor, jitter: Tensor) -> Tensor:
        """World positions of cells ``indices`` of level ``lvl`` displaced by ``jitter`` in [0, 1)^3 cell units (ref :383-391)."""
        if not indices.is_cuda:
            unit = (self.grid_coords[indices] + jitter) / self.resolution
            return self.aabbs[lvl, :3] + unit * (self.aabbs[lvl, 3:] - self.aabbs[lvl, :3])
        n = indices.shape[0]
        x = torch.empty((n, self.DIM), dtype=torch.float32, device=indices.device)
        res = (C.c_int32 * 3)(*[int(v) for v in self.binaries.shape[1:]])
        with torch.cuda.device(indices.device):
            B.call("nfa_grid_cell_points", B.ptr(indices.contiguous()), B.ptr(jitter.contiguous()), n, res,
                   self.aabbs.data_ptr() + 24 * lvl, B.ptr(x), B.stream())
        return x

    def _ema_update(self, lvl: int, indices: Tensor, occ: Tensor, ema_decay: float) -> None:
        """``occs[cells] = max(occs[cells] * decay, occ)`` (ref :393-398), in place."""
        if not self.occs.is_cuda:
            cell_ids = lvl * self.cells_per_lvl + indices
            self.occs[cell_ids] = torch.maximum(self.occs[cell_ids] * ema_decay, occ)
            return
        n = indices.shape[0]
        occ = occ.to(torch.float32).contiguous()
        assert occ.shape == (n,), "occ_eval_fn must return one value per position"
        scratch = torch.empty(n, dtype=torch.float32, device=self.occs.device)
        with torch.cuda.device(self.occs.device):
            B.call("nfa_grid_ema_update", B.ptr(self.occs), lvl * self.cells_per_lvl, B.ptr(indices.contiguous()), n, B.ptr(occ),
                   float(ema_decay), B.ptr(scratch), B.stream())
        self._occs_mean_cache = None   # (the kernel wrote through the raw pointer: the version counter did not move)

    def _rebinarize(self, occ_thre: float) -> None:
        """``binaries = occs > clamp(mean(occs[occs >= 0]), max=occ_thre)`` (ref :403-404).

        The native pass accumulates the mean in fp64 and rounds it to fp32 once; the reference's ``torch.mean`` is an fp32
        tree reduction whose order differs between its own CPU and CUDA backends, so the threshold is defined only up to
        an ulp there too: cells whose occupancy equals the threshold to within that ulp may fall on either side (the grid
        tests compare ``binaries`` away from the threshold's edge, ``tests/test_gpu_parity.py::test_grid_update_kernels_vs_oracle``)."""
        if not self.occs.is_cuda:
            thre = torch.clamp(self.occs[self.occs >= 0].mean(), max=occ_thre)
            self.binaries = (self.occs > thre).view(self.binaries.shape)
            return
        from ..grid import _walk_supported
        dev = self.occs.device
        shape = tuple(self.binaries.shape)
        if not _walk_supported(self.binaries):   # beyond the packed walk's range: the reference's expression
            thre = torch.clamp(self.occs[self.occs >= 0].mean(), max=occ_thre)
            self.binaries = (self.occs > thre).view(self.binaries.shape)
            return
        res = (C.c_int32 * 3)(*shape[1:])
        binaries = torch.empty(shape, dtype=torch.bool, device=dev)
        bits = torch.empty(int(B.load().nfa_walk_bits_words(shape[0], res)), dtype=torch.int32, device=dev)
        scratch = torch.empty(int(B.load().nfa_grid_rebinarize_scratch_bytes()) // 8 + 1, dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            B.call("nfa_grid_rebinarize", B.ptr(self.occs), shape[0], res, float(occ_thre), B.ptr(binaries), B.ptr(bits),
                   B.ptr(scratch), B.stream())
        binaries._nfa_walk_bits = (binaries._version, bits)   # the traversal's grid copy is ready: no packing pass
        self.binaries = binaries
        self._occs_mean_cache = None


_PINNED_READ = __import__("os").environ.get("NERFACC_AMD_PINNED_READ", "1") != "0"
_PINNED_TOTALS: dict = {}


def _pinned_total(dev) -> Tensor:
    buf = _PINNED_TOTALS.get(dev.index)
    if buf is None:
        buf = _PINNED_TOTALS[dev.index] = torch.empty(1, dtype=torch.int64, pin_memory=True)
    return buf


#: kept fraction of the previous batch per (n_rays, device): decides whether the next compaction is launched before its size
#: is known to the host
_KEPT_FRACTION: dict = {}
_SPECULATE_COMPACTION = __import__("os").environ.get("NERFACC_AMD_SPECULATE_COMPACTION", "1") != "0"


def _compact(seg: SegInfo, vis: Tensor, cnts: Tensor, t_starts: Tensor, t_ends: Tensor):
    """``x[masks]`` for the sampler's three arrays in one pass (ref :216-220), given the mask and the per-ray visible counts.
    Returns None when no sample is dropped (the caller keeps its arrays: same values as the boolean-index copy).

    One device->host read (the output size).  When the previous batch of this shape dropped samples, the compaction is
    launched BEFORE that read, into arrays sized from the previous batch's kept fraction: the size travels on a side stream
    that waits for the cumsum only, the host reads it while the compaction runs, and the launches that follow queue up behind
    it -- no idle GPU between the read and the next kernel (a total above the capacity: the pass wrote nothing beyond it and is
    repeated into arrays of the right size).  When the previous batch kept everything (early termination that never
    bites), nothing is launched speculatively: a compaction that copies every sample would cost more than the read."""
    dev = t_starts.device
    n = t_starts.numel()
    key = (seg.n_rays, dev.index)
    with torch.cuda.device(dev):
        total = torch.empty(1, dtype=torch.int64, device=dev)
        out_starts = _exclusive_cumsum(cnts, total)
        frac = _KEPT_FRACTION.get(key, 1.0) if _SPECULATE_COMPACTION else 1.0
        host = _pinned_total(dev)
        ri = ts = te = None
        cap = 0

        def run(ri, ts, te, capacity):
            B.call("nfa_compact_samples", B.ptr(vis), B.ptr(t_starts), B.ptr(t_ends), B.ptr(seg.packed_info),
                   B.ptr(seg.tiles), seg.n_tiles, B.ptr(out_starts), seg.n_rays, n, B.ptr(ri), B.ptr(ts), B.ptr(te), capacity, B.stream())

        if frac < 1.0 and n > 0:
            cap = min(n, ((int(n * min(1.0, frac * 1.03)) + 8192) // 4096) * 4096)
            main = torch.cuda.current_stream()
            ready = torch.cuda.Event(); ready.record(main)
            from ..grid import _side_stream
            side = _side_stream(dev)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                host.copy_(total, non_blocking=True)
                done = torch.cuda.Event(); done.record(side)
            ri = torch.empty(cap, dtype=torch.int64, device=dev)
            ts = torch.empty(cap, dtype=torch.float32, device=dev)
            te = torch.empty(cap, dtype=torch.float32, device=dev)
            run(ri, ts, te, cap)
            done.synchronize()
            m = int(host[0])
        elif _PINNED_READ:
            # the output size through pinned host memory and an event: no staging copy kernel, no implicit device sync
            host.copy_(total, non_blocking=True)
            done = torch.cuda.Event(); done.record()
            done.synchronize()
            m = int(host[0])
        else:
            m = int(total.item())
        if n > 0:
            _KEPT_FRACTION[key] = m / n
        if m == n:  # every sample is visible: x[masks] would be a copy of x
            return None
        if cap > 0 and m <= cap:
            ri, ts, te = ri[:m], ts[:m], te[:m]
        else:
            ri = torch.empty(m, dtype=torch.int64, device=dev)
            ts = torch.empty(m, dtype=torch.float32, device=dev)
            te = torch.empty(m, dtype=torch.float32, device=dev)
            if m > 0:
                run(ri, ts, te, m)
        packed = torch.stack([out_starts, cnts], dim=-1)
    info = tag_trusted(packed, m)
    tag_ray_indices(ri, seg.n_rays, info)
    return ri, ts, te


def _meshgrid3d(res: Tensor, device: Union[torch.device, str] = "cpu") -> Tensor:
    """Integer cell coordinates of a 3-D grid, shape (rx, ry, rz, 3)."""
    assert len(res) == 3
    axes = [torch.arange(int(r), dtype=torch.long) for r in res.tolist()]
    return torch.stack(torch.meshgrid(axes, indexing="ij"), dim=-1).to(device)
