"""Proposal-network estimator (ref: nerfacc/estimators/prop_net.py).

``sampling`` (ref :37-129) resamples each ray level by level with the fused inverse-CDF kernel
(csrc/pdf.hip); ``compute_loss`` / ``_pdf_loss`` (ref :131-154, :232-256) use the native
``searchsorted``.  Optimiser/scheduler glue is plain torch, as upstream.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

try:
    from typing import Literal
except ImportError:  # pragma: no cover
    from typing_extensions import Literal

import torch
from torch import Tensor
from torch.autograd.function import once_differentiable

from .. import _backend as B
from ..data_specs import RayIntervals
from ..pdf import importance_sampling, searchsorted
from ..volrend import render_transmittance_from_density
from .base import AbstractEstimator

FUSE_CDFS = os.environ.get("NERFACC_AMD_FUSE_CDFS", "1") != "0"   # A/B switch (the tests compare both forms)
FUSE_LOSS_MEAN = os.environ.get("NERFACC_AMD_FUSE_LOSS_MEAN", "1") != "0"   # A/B switch: 0 = _pdf_loss(...).mean() through the loss array


class PropNetEstimator(AbstractEstimator):
    """Proposal network transmittance estimator ("Mip-NeRF 360").

    Args:
        optimizer: optimizer of the proposal networks (optional).
        scheduler: its learning-rate scheduler (optional).
    """

    def __init__(self, optimizer: Optional[torch.optim.Optimizer] = None,
                 scheduler: Optional[torch.optim.lr_scheduler._LRScheduler] = None) -> None:
        super().__init__()
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.prop_cache: List = []

    def capture(self, step_fn: Callable, warmup: int = 3):
        """A fixed-shape training step of this estimator (``sampling`` with ``requires_grad`` -> the user's rendering ->
        ``compute_loss`` -> ``torch.autograd.grad``, all inside ``step_fn()``) as ONE hipGraph launch: returns a
        :class:`nerfacc_amd.graphs.CapturedStep`; calling it replays the step and returns ``step_fn``'s (static) outputs,
        identical to the eager step's.  The batched path has no host synchronisation, which is what makes this possible
        (the reference's does: ``.item()`` in its scans' host code).  Extension: no counterpart upstream.  The proposal
        cache is consumed inside the step, as in the eager order ``sampling`` -> ``compute_loss``."""
        from ..graphs import CapturedStep
        return CapturedStep(step_fn, warmup=warmup)

    @torch.no_grad()
    def sampling(
        self,
        prop_sigma_fns: List[Callable],
        prop_samples: List[int],
        num_samples: int,
        n_rays: int,
        near_plane: float,
        far_plane: float,
        sampling_type: Literal["uniform", "lindisp"] = "lindisp",
        stratified: bool = False,
        requires_grad: bool = False,
    ) -> Tuple[Tensor, Tensor]:
        """Sampling with CDFs from proposal networks -> ``(t_starts, t_ends)`` of shape
        ``(n_rays, num_samples)``.  With ``requires_grad`` the proposal outputs are cached for
        :meth:`update_every_n_steps` (same contract as the reference)."""
        assert len(prop_sigma_fns) == len(prop_samples), (
            "The number of proposal networks and the number of samples should be the same.")
        cdfs = torch.cat([torch.zeros((n_rays, 1), device=self.device),
                          torch.ones((n_rays, 1), device=self.device)], dim=-1)
        intervals = RayIntervals(vals=cdfs)
        for level_fn, level_samples in zip(prop_sigma_fns, prop_samples):
            intervals, t_starts, t_ends = _resample(intervals, cdfs, level_samples, stratified, sampling_type,
                                                    near_plane, far_plane)
            with torch.set_grad_enabled(requires_grad):
                sigmas = level_fn(t_starts, t_ends)
                assert sigmas.shape == t_starts.shape
                cdfs = _cdfs_from_density(t_starts, t_ends, sigmas)
                if requires_grad:
                    self.prop_cache.append((intervals, cdfs))
        intervals, t_starts, t_ends = _resample(intervals, cdfs, num_samples, stratified, sampling_type, near_plane,
                                                far_plane)
        if requires_grad:
            self.prop_cache.append((intervals, None))
        return t_starts, t_ends

    @torch.enable_grad()
    def compute_loss(self, trans: Tensor, loss_scaler: float = 1.0) -> Tensor:
        """Proposal loss from the final transmittance ``(n_rays, num_samples)`` (ref :131-154)."""
        if len(self.prop_cache) == 0:
            return torch.zeros((), device=self.device)
        intervals, _ = self.prop_cache.pop()
        cdfs = _cdfs_from_trans(trans.detach())
        loss = 0.0
        while self.prop_cache:
            prop_intervals, prop_cdfs = self.prop_cache.pop()
            loss += _pdf_loss_mean(intervals, cdfs, prop_intervals, prop_cdfs)
        return loss * loss_scaler

    @torch.enable_grad()
    def update_every_n_steps(self, trans: Tensor, requires_grad: bool = False, loss_scaler: float = 1.0) -> float:
        """Step the proposal networks when ``requires_grad``; returns the loss value (ref :156-178)."""
        if requires_grad:
            return self._update(trans=trans, loss_scaler=loss_scaler)
        if self.scheduler is not None:
            self.scheduler.step()
        return 0.0

    @torch.enable_grad()
    def _update(self, trans: Tensor, loss_scaler: float = 1.0) -> float:
        assert len(self.prop_cache) > 0
        assert self.optimizer is not None, "No optimizer is provided."
        loss = self.compute_loss(trans, loss_scaler)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        return loss.item()


def get_proposal_requires_grad_fn(target: float = 5.0, num_steps: int = 1000) -> Callable:
    """Schedule deciding on which steps the proposal networks get gradients (ref :194-212)."""
    since_last = 0

    def proposal_requires_grad_fn(step: int) -> bool:
        nonlocal since_last
        wanted_gap = min(step / num_steps, 1.0) * target
        fire = since_last > wanted_gap
        if fire:
            since_last = 0
        since_last += 1
        return fire

    return proposal_requires_grad_fn


class _CdfsFromTrans(torch.autograd.Function):
    """``1 - cat([trans, 0], -1)`` (ref :104-107, :139-142) written into one buffer: the complement goes straight
    into the first S columns and the backward is one negation of a view, instead of cat + zeros + rsub forward and
    neg + slice copy backward (0.5 ms per cfg-3 step)."""

    @staticmethod
    def forward(ctx, trans: Tensor) -> Tensor:
        S = trans.shape[-1]
        cdfs = trans.new_empty((*trans.shape[:-1], S + 1))
        torch.sub(1.0, trans, out=cdfs[..., :S])
        cdfs[..., S] = 1.0
        return cdfs

    @staticmethod
    def backward(ctx, g: Tensor):
        return torch.neg(g[..., :-1])


class _CdfsFromDensity(torch.autograd.Function):
    """``1 - cat([T, 0], -1)`` with ``T = render_transmittance_from_density(t_starts, t_ends, sigmas)`` for batched
    ``(n_rays, S)`` rows (ref :96-107) in ONE pass of the segmented engine (``nfa_density_cdf_rows_fwd``: the CDF rows are
    written next to T), and the backward from the gradient at the CDF rows straight to the densities
    (``nfa_density_cdf_rows_bwd``) -- same arithmetic as the two-step form, without its 0.13 ms strided complement per level
    forward and 0.08 ms negated slice backward.  No gradient flows to t_starts / t_ends (they come out of a no-grad
    resampling, as upstream)."""

    @staticmethod
    def forward(ctx, t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, seg) -> Tensor:
        ts, te, sg = t_starts.contiguous(), t_ends.contiguous(), sigmas.contiguous()
        R, S = sg.shape
        trans = torch.empty_like(sg)   # (alphas are neither written nor read: only T carries a gradient here)
        cdfs = sg.new_empty((R, S + 1))
        with torch.cuda.device(sg.device):
            B.call("nfa_density_cdf_rows_fwd", B.ptr(ts), B.ptr(te), B.ptr(sg), B.ptr(seg.packed_info), B.ptr(seg.tiles),
                   seg.n_tiles, seg.n_rays, sg.numel(), S, B.ptr(trans), None, B.ptr(cdfs), B.stream())
        ctx.seg = seg
        ctx.save_for_backward(ts, te, trans)
        return cdfs

    @staticmethod
    @once_differentiable
    def backward(ctx, g: Tensor):
        ts, te, trans = ctx.saved_tensors
        seg = ctx.seg
        if not ctx.needs_input_grad[2]:
            return None, None, None, None
        g = g.contiguous()
        g_sig = torch.empty_like(trans)
        with torch.cuda.device(trans.device):
            B.call("nfa_density_cdf_rows_bwd", B.ptr(ts), B.ptr(te), B.ptr(trans), None, B.ptr(g), B.ptr(seg.packed_info),
                   B.ptr(seg.tiles), seg.n_tiles, seg.n_rays, trans.numel(), trans.shape[-1], B.ptr(g_sig), B.stream())
        return None, None, g_sig, None


def _cdfs_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor) -> Tensor:
    """CDF rows of one proposal level.  The fused pass serves 2-D float32 CUDA rows whose sample positions need no
    gradient; everything else takes the two-step form."""
    from .._segments import batched_native
    fused = (sigmas.dim() == 2 and sigmas.shape[-1] >= 1 and sigmas.numel() > 0 and not t_starts.requires_grad
             and not t_ends.requires_grad and FUSE_CDFS)
    seg = batched_native(t_starts, t_ends, sigmas) if fused else None
    if seg is None:
        trans, _ = render_transmittance_from_density(t_starts, t_ends, sigmas)
        return _cdfs_from_trans(trans)
    if sigmas.requires_grad and torch.is_grad_enabled():
        return _CdfsFromDensity.apply(t_starts, t_ends, sigmas, seg)
    return _CdfsFromDensity.forward(_NoCtx(), t_starts, t_ends, sigmas, seg)


class _NoCtx:
    """Stand-in for the autograd context when the fused function runs without a graph."""
    def save_for_backward(self, *a):
        pass


def _cdfs_from_trans(trans: Tensor) -> Tensor:
    if trans.requires_grad and torch.is_grad_enabled():
        return _CdfsFromTrans.apply(trans)
    return _CdfsFromTrans.forward(None, trans)


def _resample(intervals: RayIntervals, cdfs: Tensor, n: int, stratified: bool, sampling_type: str, t_min, t_max):
    """One proposal level's resampling and s -> t mapping (ref :89-96, :120-125).  With scalar planes the mapping is
    fused into the resampling kernel; Tensor planes take the reference's tensor expression."""
    if isinstance(t_min, Tensor) or isinstance(t_max, Tensor):
        intervals, _ = importance_sampling(intervals, cdfs, n, stratified, need_samples=False)
        t_vals = _transform_stot(sampling_type, intervals.vals, t_min, t_max)
        return intervals, t_vals[..., :-1], t_vals[..., 1:]
    intervals, _, t_starts, t_ends = importance_sampling(intervals, cdfs, n, stratified,
                                                         transform=(sampling_type, t_min, t_max), need_samples=False)
    return intervals, t_starts, t_ends


def _transform_stot(transform_type: Literal["uniform", "lindisp"], s_vals: Tensor, t_min, t_max) -> Tensor:
    """Map normalised distances s in [0,1] to metric t (ref :215-229)."""
    if transform_type == "uniform":
        return s_vals * t_max + (1 - s_vals) * t_min
    if transform_type == "lindisp":
        return 1 / (s_vals * (1 / t_max) + (1 - s_vals) * (1 / t_min))
    raise ValueError(f"Unknown transform_type: {transform_type}")


class _PdfLossBatched(torch.autograd.Function):
    """The batched branch of :func:`_pdf_loss` as one native pass forward and one backward (the reference composes
    searchsorted, two gathers and five elementwise ops, and autograd adds two scatter_adds and ~10 more)."""

    @staticmethod
    def forward(ctx, q_vals, q_cdfs, k_vals, k_cdfs, eps: float):
        ctx.set_materialize_grads(False)
        qv, qc, kv, kc = (t.contiguous() for t in (q_vals, q_cdfs, k_vals, k_cdfs))
        dev = B.require_device(qv, qc, kv, kc)
        Q1, K1 = qv.shape[-1], kv.shape[-1]
        n_rays = qv.numel() // Q1
        loss = torch.empty(qv.shape[:-1] + (Q1 - 1,), dtype=torch.float32, device=dev)
        need_bwd = any(ctx.needs_input_grad)
        ids = torch.empty(loss.shape, dtype=torch.int32, device=dev) if need_bwd else None  # key edges, for the backward
        with torch.cuda.device(dev):
            B.call("nfa_pdf_loss_fwd", B.ptr(qv), B.ptr(qc), B.ptr(kv), B.ptr(kc), n_rays, Q1, K1, float(eps), B.ptr(loss),
                   B.ptr(ids), B.stream())
        if need_bwd:
            ctx.save_for_backward(qc, kc, ids)
        ctx.eps, ctx.K1 = float(eps), K1
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loss):
        if g_loss is None or not (ctx.needs_input_grad[1] or ctx.needs_input_grad[3]):
            return None, None, None, None, None
        qc, kc, ids = ctx.saved_tensors
        Q1, K1 = qc.shape[-1], kc.shape[-1]
        n_rays = qc.numel() // Q1
        g_kc = torch.empty_like(kc)
        g_qc = torch.empty_like(qc) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(qc.device):
            B.call("nfa_pdf_loss_bwd", B.ptr(qc), B.ptr(kc), B.ptr(ids), n_rays, Q1, K1, ctx.eps,
                   B.ptr(g_loss.contiguous()), B.ptr(g_kc), B.ptr(g_qc), B.stream())
        return None, g_qc, None, (g_kc if ctx.needs_input_grad[3] else None), None


class _PdfLossBatchedMean(torch.autograd.Function):
    """``_pdf_loss(...).mean()`` of the batched branch without the loss array: the forward pass leaves one partial sum per wave
    (added up here: a few thousand floats), the backward pass takes the mean's scalar gradient -- the (R, S) loss, its
    reduction and the expanded (R, S) gradient never touch memory (ref: prop_net.py:151, :232-256)."""

    @staticmethod
    def forward(ctx, q_vals, q_cdfs, k_vals, k_cdfs, eps: float):
        ctx.set_materialize_grads(False)
        qv, qc, kv, kc = (t.contiguous() for t in (q_vals, q_cdfs, k_vals, k_cdfs))
        dev = B.require_device(qv, qc, kv, kc)
        Q1, K1 = qv.shape[-1], kv.shape[-1]
        n_rays = qv.numel() // Q1
        need_bwd = any(ctx.needs_input_grad)
        with torch.cuda.device(dev):
            n_part = int(B.load().nfa_pdf_loss_partials(n_rays, Q1, K1))
            partials = torch.empty(max(n_part, 1), dtype=torch.float32, device=dev)
            ids = torch.empty(qv.shape[:-1] + (Q1 - 1,), dtype=torch.int32, device=dev) if need_bwd else None
            if n_part == 0:
                partials.zero_()
            B.call("nfa_pdf_loss_sum_fwd", B.ptr(qv), B.ptr(qc), B.ptr(kv), B.ptr(kc), n_rays, Q1, K1, float(eps), B.ptr(partials),
                   B.ptr(ids), B.stream())
        if need_bwd:
            ctx.save_for_backward(qc, kc, ids)
        ctx.eps = float(eps)
        return partials.sum() / float(max(n_rays * (Q1 - 1), 1))

    @staticmethod
    @once_differentiable
    def backward(ctx, g_mean):
        if g_mean is None or not (ctx.needs_input_grad[1] or ctx.needs_input_grad[3]):
            return None, None, None, None, None
        qc, kc, ids = ctx.saved_tensors
        Q1, K1 = qc.shape[-1], kc.shape[-1]
        n_rays = qc.numel() // Q1
        g_kc = torch.empty_like(kc)
        g_qc = torch.empty_like(qc) if ctx.needs_input_grad[1] else None
        g = g_mean.to(torch.float32).reshape(1).contiguous()
        with torch.cuda.device(qc.device):
            B.call("nfa_pdf_loss_mean_bwd", B.ptr(qc), B.ptr(kc), B.ptr(ids), n_rays, Q1, K1, ctx.eps, B.ptr(g), B.ptr(g_kc), B.ptr(g_qc),
                   B.stream())
        return None, g_qc, None, (g_kc if ctx.needs_input_grad[3] else None), None


def _pdf_loss_mean(segments_query: RayIntervals, cdfs_query: Tensor, segments_key: RayIntervals, cdfs_key: Tensor,
                   eps: float = 1e-7) -> Tensor:
    """``_pdf_loss(...).mean()`` (what ``compute_loss`` adds up, ref :151); batched float32 CUDA rows take the form that
    never materialises the per-interval loss."""
    qv, kv = segments_query.vals, segments_key.vals
    if (FUSE_LOSS_MEAN and qv.dim() > 1 and kv.dim() > 1 and qv.is_cuda
            and all(t.dtype == torch.float32 for t in (qv, kv, cdfs_query, cdfs_key))
            and qv.shape[:-1] == kv.shape[:-1] and cdfs_query.shape == qv.shape and cdfs_key.shape == kv.shape
            and 2 <= qv.shape[-1] <= 1024 and kv.shape[-1] <= 1024 and qv.numel() > 0):
        return _PdfLossBatchedMean.apply(qv, cdfs_query, kv, cdfs_key, eps)
    return _pdf_loss(segments_query, cdfs_query, segments_key, cdfs_key, eps).mean()


def _pdf_loss(segments_query: RayIntervals, cdfs_query: Tensor, segments_key: RayIntervals, cdfs_key: Tensor,
              eps: float = 1e-7) -> Tensor:
    """Mip-NeRF-360 interlevel loss ``clip(w - w_outer, 0)^2 / (w + eps)`` (ref :232-256)."""
    qv, kv = segments_query.vals, segments_key.vals
    if (qv.dim() > 1 and kv.dim() > 1 and qv.is_cuda and all(t.dtype == torch.float32 for t in (qv, kv, cdfs_query, cdfs_key))
            and qv.shape[:-1] == kv.shape[:-1] and cdfs_query.shape == qv.shape and cdfs_key.shape == kv.shape
            and 2 <= qv.shape[-1] <= 1024 and kv.shape[-1] <= 1024):
        return _PdfLossBatched.apply(qv, cdfs_query, kv, cdfs_key, eps)
    ids_left, ids_right = searchsorted(segments_key, segments_query)
    if segments_query.vals.dim() > 1:
        w = cdfs_query[..., 1:] - cdfs_query[..., :-1]
        ids_left = ids_left[..., :-1]
        ids_right = ids_right[..., 1:]
    else:
        assert segments_query.is_left is not None and segments_query.is_right is not None
        w = cdfs_query[segments_query.is_right] - cdfs_query[segments_query.is_left]
        ids_left = ids_left[segments_query.is_left]
        ids_right = ids_right[segments_query.is_right]
    w_outer = cdfs_key.gather(-1, ids_right) - cdfs_key.gather(-1, ids_left)
    return torch.clip(w - w_outer, min=0) ** 2 / (w + eps)
