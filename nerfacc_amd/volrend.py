"""Volumetric rendering ops (ref: nerfacc/volrend.py).

Public functions, argument meaning, asserts and return values follow the reference
(rendering :14-158, render_transmittance_* :161-264, render_weight_* :267-362,
render_visibility_* :365-480, accumulate_along_rays(_) :483-573).  For flattened inputs the
reference expands each op into ~8 elementwise ATen launches around one scan kernel; here each op
is ONE fused pass of the segmented engine (csrc/segscan.hip) with a hand-written backward.
Batched inputs (no packed_info / ray_indices) take the same pure-torch route as the reference.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch
from torch import Tensor
from torch.autograd.function import once_differentiable

from . import _backend as B
from ._segments import SegInfo, batched_native, resolve, seginfo_from_ray_indices
from .scan import exclusive_prod, exclusive_sum


# rendering(): one fused pass each way (True) or weights + accumulation as two passes (False, for A/B tests)
FUSE_RENDERING = True


def _f32c(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError("nerfacc_amd: packed rendering ops support float32 only")
    return t.contiguous()


# --------------------------------------------------------------------------- fused packed ops
class _RenderFromDensity(torch.autograd.Function):
    """(weights, trans, alphas) from (t_starts, t_ends, sigmas[, prefix_trans]) in one pass."""

    @staticmethod
    def forward(ctx, t_starts, t_ends, sigmas, prefix_trans, seg: SegInfo, want_weights: bool):
        ctx.set_materialize_grads(False)  # unused outputs arrive as None, not as zero tensors
        ts, te, sg, pf = _f32c(t_starts), _f32c(t_ends), _f32c(sigmas), _f32c(prefix_trans)
        dev = B.require_device(ts, te, sg, pf)
        n = sg.numel()
        weights = torch.empty_like(sg) if want_weights else None
        trans, alphas = torch.empty_like(sg), torch.empty_like(sg)
        if n:
            with torch.cuda.device(dev):
                B.call("nfa_render_from_density_fwd", B.ptr(ts), B.ptr(te), B.ptr(sg), B.ptr(pf),
                       B.ptr(seg.packed_info), B.ptr(seg.tiles), seg.n_tiles, seg.n_rays, n, B.ptr(weights), B.ptr(trans),
                       B.ptr(alphas), B.stream())
        ctx.seg = seg
        ctx.want_weights = want_weights
        ctx.save_for_backward(ts, te, sg, trans, alphas)
        if not want_weights:
            weights = trans.new_empty(0)
            ctx.mark_non_differentiable(weights)
        return weights, trans, alphas

    @staticmethod
    @once_differentiable
    def backward(ctx, g_w, g_t, g_a):
        ts, te, sg, trans, alphas = ctx.saved_tensors
        seg = ctx.seg
        n = sg.numel()
        need_ts, need_te, need_sg = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if ctx.needs_input_grad[3]:
            raise NotImplementedError("nerfacc_amd: gradient w.r.t. prefix_trans is not implemented")
        g_w = None if (g_w is None or not ctx.want_weights) else _f32c(g_w)
        g_t = None if g_t is None else _f32c(g_t)
        g_a = None if g_a is None else _f32c(g_a)
        need_x = need_ts or need_te
        g_sig = torch.empty_like(sg) if need_sg else None
        g_x = torch.empty_like(sg) if need_x else None
        if n and (need_sg or need_x):
            with torch.cuda.device(sg.device):
                B.call("nfa_render_from_density_bwd", B.ptr(ts), B.ptr(te), B.ptr(trans), B.ptr(alphas), B.ptr(g_w),
                       B.ptr(g_t), B.ptr(g_a), B.ptr(seg.packed_info), B.ptr(seg.tiles), seg.n_tiles, seg.n_rays, n,
                       B.ptr(g_sig), B.ptr(g_x), B.stream())
        g_ts = (-(g_x * sg)) if need_ts else None
        g_te = (g_x * sg) if need_te else None
        return g_ts, g_te, g_sig, None, None, None


class _RenderFromAlpha(torch.autograd.Function):
    """(weights, trans) from alphas[, prefix_trans] in one pass."""

    @staticmethod
    def forward(ctx, alphas, prefix_trans, seg: SegInfo, want_weights: bool):
        ctx.set_materialize_grads(False)  # unused outputs arrive as None, not as zero tensors
        al, pf = _f32c(alphas), _f32c(prefix_trans)
        dev = B.require_device(al, pf)
        n = al.numel()
        weights = torch.empty_like(al) if want_weights else None
        trans = torch.empty_like(al)
        if n:
            with torch.cuda.device(dev):
                B.call("nfa_render_from_alpha_fwd", B.ptr(al), B.ptr(pf), B.ptr(seg.packed_info), B.ptr(seg.tiles), seg.n_tiles,
                       seg.n_rays, n, B.ptr(weights), B.ptr(trans), B.stream())
        ctx.seg, ctx.want_weights = seg, want_weights
        ctx.save_for_backward(al, trans)
        if not want_weights:
            weights = trans.new_empty(0)
            ctx.mark_non_differentiable(weights)
        return weights, trans

    @staticmethod
    @once_differentiable
    def backward(ctx, g_w, g_t):
        al, trans = ctx.saved_tensors
        seg = ctx.seg
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("nerfacc_amd: gradient w.r.t. prefix_trans is not implemented")
        if not ctx.needs_input_grad[0]:
            return None, None, None, None
        g_w = None if (g_w is None or not ctx.want_weights) else _f32c(g_w)
        g_t = None if g_t is None else _f32c(g_t)
        g_al = torch.empty_like(al)
        if al.numel():
            with torch.cuda.device(al.device):
                B.call("nfa_render_from_alpha_bwd", B.ptr(al), B.ptr(trans), B.ptr(g_w), B.ptr(g_t),
                       B.ptr(seg.packed_info), B.ptr(seg.tiles), seg.n_tiles, seg.n_rays, al.numel(), B.ptr(g_al), B.stream())
        return g_al, None, None, None


class _Accumulate(torch.autograd.Function):
    """out[r] = sum_{i in ray r} w_i * values_i  (deterministic segmented reduction)."""

    @staticmethod
    def forward(ctx, weights, values, seg: SegInfo):
        ctx.set_materialize_grads(False)  # unused outputs arrive as None, not as zero tensors
        w, v = _f32c(weights), _f32c(values)
        dev = B.require_device(w, v)
        D = 1 if v is None else v.shape[-1]
        out = torch.empty((seg.n_rays, D), dtype=torch.float32, device=dev)
        if seg.n_rays and D:
            with torch.cuda.device(dev):
                B.call("nfa_accumulate_along_rays", B.ptr(w), B.ptr(v), D, B.ptr(seg.packed_info), B.ptr(seg.tiles), seg.n_tiles,
                       seg.n_rays, w.numel(), 0, B.ptr(out), B.stream())
        ctx.seg, ctx.D, ctx.has_values = seg, D, v is not None
        ctx.save_for_backward(w, v if v is not None else w.new_empty(0))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out):
        w, v = ctx.saved_tensors
        v = v if ctx.has_values else None
        seg = ctx.seg
        g_out = _f32c(g_out)
        need_w, need_v = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and v is not None
        g_w = torch.empty_like(w) if need_w else None
        g_v = torch.empty_like(v) if need_v else None
        if w.numel() and (need_w or need_v):
            with torch.cuda.device(w.device):
                B.call("nfa_accumulate_along_rays_bwd", B.ptr(w), B.ptr(v), ctx.D, B.ptr(g_out),
                       B.ptr(seg.packed_info), B.ptr(seg.tiles), seg.n_tiles, seg.n_rays, w.numel(), B.ptr(g_w), B.ptr(g_v),
                       B.stream())
        return g_w, g_v, None


class _RenderAccumulate(torch.autograd.Function):
    """The three accumulations of ``rendering`` (colours, opacity, un-normalised depth) fused."""

    @staticmethod
    def forward(ctx, weights, rgbs, t_starts, t_ends, seg: SegInfo):
        ctx.set_materialize_grads(False)  # unused outputs arrive as None, not as zero tensors
        w, c, ts, te = _f32c(weights), _f32c(rgbs), _f32c(t_starts), _f32c(t_ends)
        dev = B.require_device(w, c, ts, te)
        R = seg.n_rays
        colors = torch.empty((R, 3), dtype=torch.float32, device=dev)
        opac = torch.empty((R, 1), dtype=torch.float32, device=dev)
        depth = torch.empty((R, 1), dtype=torch.float32, device=dev)
        if R:
            with torch.cuda.device(dev):
                B.call("nfa_render_accumulate_fwd", B.ptr(w), B.ptr(c), B.ptr(ts), B.ptr(te), B.ptr(seg.packed_info),
                       B.ptr(seg.tiles), seg.n_tiles, R, w.numel(), B.ptr(colors), B.ptr(opac), B.ptr(depth), B.stream())
        ctx.seg = seg
        ctx.save_for_backward(w, c, ts, te)
        return colors, opac, depth

    @staticmethod
    @once_differentiable
    def backward(ctx, g_c, g_o, g_d):
        w, c, ts, te = ctx.saved_tensors
        seg = ctx.seg
        need_w, need_c = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_w = torch.empty_like(w) if need_w else None
        g_rgb = torch.empty_like(c) if need_c else None
        if w.numel() and (need_w or need_c):
            with torch.cuda.device(w.device):
                B.call("nfa_render_accumulate_bwd", B.ptr(w), B.ptr(c), B.ptr(ts), B.ptr(te), B.ptr(_f32c(g_c)),
                       B.ptr(_f32c(g_o)), B.ptr(_f32c(g_d)), B.ptr(seg.packed_info), B.ptr(seg.tiles), seg.n_tiles, seg.n_rays,
                       w.numel(), B.ptr(g_w), B.ptr(g_rgb), B.stream())
        return g_w, g_rgb, None, None, None


class _RenderFused(torch.autograd.Function):
    """``rendering`` with a density callback as one forward and one backward pass over the samples.

    Replaces render_weight_from_density + the three accumulations (volrend.py:109-151): the reference
    runs ~20 ATen kernels there; the unfused native path two each way.  Outputs are bit-identical to
    ``_RenderFromDensity`` followed by ``_RenderAccumulate``.
    """

    @staticmethod
    def forward(ctx, t_starts, t_ends, sigmas, rgbs, seg: SegInfo):
        ctx.set_materialize_grads(False)  # unused outputs arrive as None, not as zero tensors
        ts, te, sg, c = _f32c(t_starts), _f32c(t_ends), _f32c(sigmas), _f32c(rgbs)
        dev = B.require_device(ts, te, sg, c)
        R, n = seg.n_rays, sg.numel()
        weights, trans, alphas = torch.empty_like(sg), torch.empty_like(sg), torch.empty_like(sg)
        colors = torch.empty((R, 3), dtype=torch.float32, device=dev)
        opac = torch.empty((R, 1), dtype=torch.float32, device=dev)
        depth = torch.empty((R, 1), dtype=torch.float32, device=dev)
        if R:
            with torch.cuda.device(dev):
                B.call("nfa_render_fused_fwd", B.ptr(ts), B.ptr(te), B.ptr(sg), B.ptr(c), B.ptr(seg.packed_info),
                       B.ptr(seg.tiles), seg.n_tiles, R, n, B.ptr(weights), B.ptr(trans), B.ptr(alphas), B.ptr(colors),
                       B.ptr(opac), B.ptr(depth), B.stream())
        ctx.seg = seg
        ctx.save_for_backward(ts, te, c, trans, alphas)
        return colors, opac, depth, weights, trans, alphas

    @staticmethod
    @once_differentiable
    def backward(ctx, g_c, g_o, g_d, g_w, g_t, g_a):
        ts, te, c, trans, alphas = ctx.saved_tensors
        seg = ctx.seg
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            raise NotImplementedError("nerfacc_amd: rendering is not differentiable w.r.t. t_starts / t_ends "
                                      "(same contract as the reference, volrend.py:33-35)")
        need_sg, need_c = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        g_sig = torch.empty_like(trans) if need_sg else None
        g_rgb = torch.empty_like(c) if need_c else None
        if trans.numel() and (need_sg or need_c):
            with torch.cuda.device(trans.device):
                B.call("nfa_render_fused_bwd", B.ptr(ts), B.ptr(te), B.ptr(c), B.ptr(trans), B.ptr(alphas),
                       B.ptr(_f32c(g_c)), B.ptr(_f32c(g_o)), B.ptr(_f32c(g_d)), B.ptr(_f32c(g_w)), B.ptr(_f32c(g_t)),
                       B.ptr(_f32c(g_a)), B.ptr(seg.packed_info), B.ptr(seg.tiles), seg.n_tiles, seg.n_rays,
                       trans.numel(), B.ptr(g_sig), B.ptr(g_rgb), B.stream())
        return None, None, g_sig, g_rgb, None


def _use_fused(seg: Optional[SegInfo], *tensors: Optional[Tensor]) -> bool:
    if seg is None or not seg.contiguous:
        return False
    return all(t is None or (t.dim() == 1 and t.dtype == torch.float32) for t in tensors)


def _prefix_needs_grad(prefix_trans: Optional[Tensor]) -> bool:
    return prefix_trans is not None and prefix_trans.requires_grad and torch.is_grad_enabled()


# --------------------------------------------------------------------------- public API
def rendering(
    t_starts: Tensor,
    t_ends: Tensor,
    ray_indices: Optional[Tensor] = None,
    n_rays: Optional[int] = None,
    rgb_sigma_fn: Optional[Callable] = None,
    rgb_alpha_fn: Optional[Callable] = None,
    render_bkgd: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor, Tensor, Dict]:
    """Render rays through the radiance field defined by ``rgb_sigma_fn`` / ``rgb_alpha_fn``.

    Same contract as the reference (volrend.py:14-158): differentiable to the callback's
    outputs, not to ``t_starts`` / ``t_ends`` / ``ray_indices``; returns
    ``(colors (n_rays,3), opacities (n_rays,1), depths (n_rays,1), extras)``.
    """
    if ray_indices is not None:
        assert (
            t_starts.shape == t_ends.shape == ray_indices.shape
        ), "Since nerfacc 0.5.0, t_starts, t_ends and ray_indices must have the same shape (N,). "
    if rgb_sigma_fn is None and rgb_alpha_fn is None:
        raise ValueError("At least one of `rgb_sigma_fn` and `rgb_alpha_fn` should be specified.")

    seg = None
    if ray_indices is not None and ray_indices.is_cuda:
        assert n_rays is not None, "n_rays must be provided"
        seg = seginfo_from_ray_indices(ray_indices, n_rays)

    if rgb_sigma_fn is not None:
        if t_starts.shape[0] != 0:
            rgbs, sigmas = rgb_sigma_fn(t_starts, t_ends, ray_indices)
        else:
            rgbs = torch.empty((0, 3), device=t_starts.device)
            sigmas = torch.empty((0,), device=t_starts.device)
        assert rgbs.shape[-1] == 3, "rgbs must have 3 channels, got {}".format(rgbs.shape)
        assert sigmas.shape == t_starts.shape, "sigmas must have shape of (N,)! Got {}".format(sigmas.shape)
        if FUSE_RENDERING and seg is not None and seg.sorted_indices and _use_fused(seg, sigmas, t_starts, t_ends) \
                and rgbs.dtype == torch.float32 and rgbs.dim() == 2 \
                and not (t_starts.requires_grad or t_ends.requires_grad):
            # the whole of volrend.py:109-151 as one pass each way
            colors, opacities, depths, weights, trans, alphas = _RenderFused.apply(t_starts, t_ends, sigmas, rgbs, seg)
            extras = {"weights": weights, "alphas": alphas, "trans": trans, "sigmas": sigmas, "rgbs": rgbs}
            return _finish_rendering(colors, opacities, depths, extras, rgbs, render_bkgd)
        weights, trans, alphas = render_weight_from_density(
            t_starts, t_ends, sigmas, ray_indices=ray_indices, n_rays=n_rays
        )
        extras = {"weights": weights, "alphas": alphas, "trans": trans, "sigmas": sigmas, "rgbs": rgbs}
    else:
        if t_starts.shape[0] != 0:
            rgbs, alphas = rgb_alpha_fn(t_starts, t_ends, ray_indices)
        else:
            rgbs = torch.empty((0, 3), device=t_starts.device)
            alphas = torch.empty((0,), device=t_starts.device)
        assert rgbs.shape[-1] == 3, "rgbs must have 3 channels, got {}".format(rgbs.shape)
        assert alphas.shape == t_starts.shape, "alphas must have shape of (N,)! Got {}".format(alphas.shape)
        weights, trans = render_weight_from_alpha(alphas, ray_indices=ray_indices, n_rays=n_rays)
        extras = {"weights": weights, "trans": trans, "rgbs": rgbs, "alphas": alphas}

    if seg is not None and seg.sorted_indices and _use_fused(seg, weights, t_starts, t_ends) \
            and rgbs.dtype == torch.float32:
        # one pass for colours, opacity and depth (the reference runs 3 x (mul + index_add_))
        colors, opacities, depths = _RenderAccumulate.apply(weights, rgbs, t_starts, t_ends, seg)
    else:
        colors = accumulate_along_rays(weights, values=rgbs, ray_indices=ray_indices, n_rays=n_rays)
        opacities = accumulate_along_rays(weights, values=None, ray_indices=ray_indices, n_rays=n_rays)
        depths = accumulate_along_rays(
            weights, values=(t_starts + t_ends)[..., None] / 2.0, ray_indices=ray_indices, n_rays=n_rays
        )
    return _finish_rendering(colors, opacities, depths, extras, rgbs, render_bkgd)


def _finish_rendering(colors, opacities, depths, extras, rgbs, render_bkgd):
    # ref: volrend.py:152-158
    depths = depths / opacities.clamp_min(torch.finfo(rgbs.dtype).eps)
    if render_bkgd is not None:
        colors = colors + render_bkgd * (1.0 - opacities)
    return colors, opacities, depths, extras


def _batched_as_packed(fn, tensors, packed_info, ray_indices, **kw):
    """Batched ``(..., S)`` CUDA float32 inputs: run the packed native op on the flat view with uniform
    segments (the reference composes torch.cumsum / cumprod here, volrend.py:203-206,259-264) and give
    the outputs the batched shape back.  Returns None when the torch composition must be used."""
    if packed_info is not None or ray_indices is not None:
        return None
    useg = batched_native(*tensors)
    if useg is None:
        return None
    shape = next(t for t in tensors if t is not None).shape
    flat = [None if t is None else t.contiguous().view(-1) for t in tensors]
    out = fn(*flat[:-1], packed_info=useg.packed_info, prefix_trans=flat[-1], **kw)
    return tuple(o.view(shape) for o in out) if isinstance(out, tuple) else out.view(shape)


def render_transmittance_from_alpha(
    alphas: Tensor,
    packed_info: Optional[Tensor] = None,
    ray_indices: Optional[Tensor] = None,
    n_rays: Optional[int] = None,
    prefix_trans: Optional[Tensor] = None,
) -> Tensor:
    """Transmittance ``T_i = prod_{j<i}(1 - alpha_j)`` (ref: volrend.py:161-206)."""
    out = _batched_as_packed(render_transmittance_from_alpha, (alphas, prefix_trans), packed_info, ray_indices)
    if out is not None:
        return out
    seg = resolve(alphas.numel(), packed_info, ray_indices, n_rays) if alphas.dim() == 1 else None
    if _use_fused(seg, alphas, prefix_trans) and not _prefix_needs_grad(prefix_trans):
        _, trans = _RenderFromAlpha.apply(alphas, prefix_trans, seg, False)
        return trans
    trans = exclusive_prod(1 - alphas, seg.packed_info if seg is not None else None)
    if prefix_trans is not None:
        trans = trans * prefix_trans
    return trans


def render_transmittance_from_density(
    t_starts: Tensor,
    t_ends: Tensor,
    sigmas: Tensor,
    packed_info: Optional[Tensor] = None,
    ray_indices: Optional[Tensor] = None,
    n_rays: Optional[int] = None,
    prefix_trans: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor]:
    """``T_i = exp(-sum_{j<i} sigma_j delta_j)`` and ``alpha_i`` (ref: volrend.py:209-264)."""
    out = _batched_as_packed(render_transmittance_from_density, (t_starts, t_ends, sigmas, prefix_trans), packed_info, ray_indices)
    if out is not None:
        return out
    seg = resolve(sigmas.numel(), packed_info, ray_indices, n_rays) if sigmas.dim() == 1 else None
    if _use_fused(seg, t_starts, t_ends, sigmas, prefix_trans) and not _prefix_needs_grad(prefix_trans):
        _, trans, alphas = _RenderFromDensity.apply(t_starts, t_ends, sigmas, prefix_trans, seg, False)
        return trans, alphas
    sigmas_dt = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sigmas_dt)
    trans = torch.exp(-exclusive_sum(sigmas_dt, seg.packed_info if seg is not None else None))
    if prefix_trans is not None:
        trans = trans * prefix_trans
    return trans, alphas


def render_weight_from_alpha(
    alphas: Tensor,
    packed_info: Optional[Tensor] = None,
    ray_indices: Optional[Tensor] = None,
    n_rays: Optional[int] = None,
    prefix_trans: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor]:
    """``w_i = T_i alpha_i``; returns (weights, transmittance) (ref: volrend.py:267-309)."""
    out = _batched_as_packed(render_weight_from_alpha, (alphas, prefix_trans), packed_info, ray_indices)
    if out is not None:
        return out
    seg = resolve(alphas.numel(), packed_info, ray_indices, n_rays) if alphas.dim() == 1 else None
    if _use_fused(seg, alphas, prefix_trans) and not _prefix_needs_grad(prefix_trans):
        return _RenderFromAlpha.apply(alphas, prefix_trans, seg, True)
    trans = render_transmittance_from_alpha(alphas, seg.packed_info if seg is not None else None,
                                            None, None, prefix_trans)
    return trans * alphas, trans


def render_weight_from_density(
    t_starts: Tensor,
    t_ends: Tensor,
    sigmas: Tensor,
    packed_info: Optional[Tensor] = None,
    ray_indices: Optional[Tensor] = None,
    n_rays: Optional[int] = None,
    prefix_trans: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor, Tensor]:
    """``w_i = T_i (1 - exp(-sigma_i delta_i))``; returns (weights, transmittance, alphas)
    (ref: volrend.py:312-362)."""
    out = _batched_as_packed(render_weight_from_density, (t_starts, t_ends, sigmas, prefix_trans), packed_info, ray_indices)
    if out is not None:
        return out
    seg = resolve(sigmas.numel(), packed_info, ray_indices, n_rays) if sigmas.dim() == 1 else None
    if _use_fused(seg, t_starts, t_ends, sigmas, prefix_trans) and not _prefix_needs_grad(prefix_trans):
        return _RenderFromDensity.apply(t_starts, t_ends, sigmas, prefix_trans, seg, True)
    trans, alphas = render_transmittance_from_density(
        t_starts, t_ends, sigmas, seg.packed_info if seg is not None else None, None, None, prefix_trans)
    return trans * alphas, trans, alphas


def _visibility_native(seg: SegInfo, t_starts, t_ends, vals, prefix_trans, early_stop_eps, alpha_thre,
                       want_counts: bool = False):
    ts, te, v, pf = _f32c(t_starts), _f32c(t_ends), _f32c(vals), _f32c(prefix_trans)
    dev = B.require_device(v, ts, te, pf)
    n = v.numel()
    vis = torch.empty(n, dtype=torch.bool, device=dev)
    cnts = torch.empty(seg.n_rays, dtype=torch.int64, device=dev) if want_counts else None
    with torch.cuda.device(dev):
        B.call("nfa_render_visibility", B.ptr(ts), B.ptr(te), B.ptr(v), B.ptr(pf), float(early_stop_eps),
               float(alpha_thre), B.ptr(seg.packed_info), B.ptr(seg.tiles), seg.n_tiles, seg.n_rays, n, B.ptr(vis), B.ptr(cnts),
               B.stream())
    return (vis, cnts) if want_counts else vis


@torch.no_grad()
def render_visibility_from_alpha(
    alphas: Tensor,
    packed_info: Optional[Tensor] = None,
    ray_indices: Optional[Tensor] = None,
    n_rays: Optional[int] = None,
    early_stop_eps: float = 1e-4,
    alpha_thre: float = 0.0,
    prefix_trans: Optional[Tensor] = None,
) -> Tensor:
    """Visibility mask ``T >= early_stop_eps`` and, if ``alpha_thre > 0``, ``alpha >= alpha_thre``
    (ref: volrend.py:365-418)."""
    out = _batched_as_packed(render_visibility_from_alpha, (alphas, prefix_trans), packed_info, ray_indices, early_stop_eps=early_stop_eps, alpha_thre=alpha_thre)
    if out is not None:
        return out
    seg = resolve(alphas.numel(), packed_info, ray_indices, n_rays) if alphas.dim() == 1 else None
    if _use_fused(seg, alphas, prefix_trans):
        return _visibility_native(seg, None, None, alphas, prefix_trans, early_stop_eps, alpha_thre)
    trans = render_transmittance_from_alpha(alphas, seg.packed_info if seg is not None else None, None, None,
                                            prefix_trans)
    vis = trans >= early_stop_eps
    if alpha_thre > 0:
        vis = vis & (alphas >= alpha_thre)
    return vis


@torch.no_grad()
def render_visibility_from_density(
    t_starts: Tensor,
    t_ends: Tensor,
    sigmas: Tensor,
    packed_info: Optional[Tensor] = None,
    ray_indices: Optional[Tensor] = None,
    n_rays: Optional[int] = None,
    early_stop_eps: float = 1e-4,
    alpha_thre: float = 0.0,
    prefix_trans: Optional[Tensor] = None,
) -> Tensor:
    """Visibility mask from densities (ref: volrend.py:421-480)."""
    out = _batched_as_packed(render_visibility_from_density, (t_starts, t_ends, sigmas, prefix_trans), packed_info, ray_indices, early_stop_eps=early_stop_eps, alpha_thre=alpha_thre)
    if out is not None:
        return out
    seg = resolve(sigmas.numel(), packed_info, ray_indices, n_rays) if sigmas.dim() == 1 else None
    if _use_fused(seg, t_starts, t_ends, sigmas, prefix_trans):
        return _visibility_native(seg, t_starts, t_ends, sigmas, prefix_trans, early_stop_eps, alpha_thre)
    trans, alphas = render_transmittance_from_density(
        t_starts, t_ends, sigmas, seg.packed_info if seg is not None else None, None, None, prefix_trans)
    vis = trans >= early_stop_eps
    if alpha_thre > 0:
        vis = vis & (alphas >= alpha_thre)
    return vis


def accumulate_along_rays(
    weights: Tensor,
    values: Optional[Tensor] = None,
    ray_indices: Optional[Tensor] = None,
    n_rays: Optional[int] = None,
) -> Tensor:
    """Accumulate ``weights * values`` along each ray -> (n_rays, D) (ref: volrend.py:483-547).

    Flattened inputs with ray-sorted ``ray_indices`` (what ``sampling`` returns) use a
    deterministic segmented reduction instead of the reference's float atomics
    (``index_add_``); unsorted indices fall back to ``index_add_`` semantics.
    """
    if values is not None:
        assert values.dim() == weights.dim() + 1
        assert weights.shape == values.shape[:-1]
    if ray_indices is not None:
        assert n_rays is not None, "n_rays must be provided"
        assert weights.dim() == 1, "weights must be flattened"
        if weights.is_cuda and weights.dtype == torch.float32 and (values is None or values.dtype == torch.float32):
            seg = seginfo_from_ray_indices(ray_indices, n_rays)
            if seg.sorted_indices:
                return _Accumulate.apply(weights, values, seg)
        src = weights[..., None] if values is None else weights[..., None] * values
        outputs = torch.zeros((n_rays, src.shape[-1]), device=src.device, dtype=src.dtype)
        outputs.index_add_(0, ray_indices, src)
        return outputs
    src = weights[..., None] if values is None else weights[..., None] * values
    return torch.sum(src, dim=-2)


def accumulate_along_rays_(
    weights: Tensor,
    values: Optional[Tensor] = None,
    ray_indices: Optional[Tensor] = None,
    outputs: Optional[Tensor] = None,
) -> None:
    """In-place version of :func:`accumulate_along_rays` (ref: volrend.py:550-573)."""
    if values is not None:
        assert values.dim() == weights.dim() + 1
        assert weights.shape == values.shape[:-1]
    if ray_indices is not None:
        assert weights.dim() == 1, "weights must be flattened"
        D = 1 if values is None else values.shape[-1]
        assert outputs.dim() == 2 and outputs.shape[-1] == D, "outputs must be of shape (n_rays, D)"
        native = (weights.is_cuda and weights.dtype == torch.float32 and outputs.dtype == torch.float32
                  and outputs.is_contiguous() and (values is None or values.dtype == torch.float32)
                  and not (torch.is_grad_enabled() and (weights.requires_grad or
                                                        (values is not None and values.requires_grad))))
        if native:
            n_rays = outputs.shape[0]
            seg = seginfo_from_ray_indices(ray_indices, n_rays)
            w, v = _f32c(weights), _f32c(values)
            with torch.cuda.device(w.device):
                if seg.sorted_indices:
                    B.call("nfa_accumulate_along_rays", B.ptr(w), B.ptr(v), D, B.ptr(seg.packed_info),
                           B.ptr(seg.tiles), seg.n_tiles, n_rays, w.numel(), 1, B.ptr(outputs), B.stream())
                else:
                    ri = ray_indices.to(torch.int64).contiguous()
                    B.call("nfa_accumulate_along_rays_atomic", B.ptr(w), B.ptr(v), D, B.ptr(ri), n_rays, w.numel(),
                           B.ptr(outputs), B.stream())
            return
        src = weights[..., None] if values is None else weights[..., None] * values
        outputs.index_add_(0, ray_indices, src)
    else:
        src = weights[..., None] if values is None else weights[..., None] * values
        outputs.add_(src.sum(dim=-2))
