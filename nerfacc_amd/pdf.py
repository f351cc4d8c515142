"""Per-ray ``searchsorted`` and inverse-CDF ``importance_sampling`` (ref: nerfacc/pdf.py)."""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import Tensor

from . import _backend as B
from .data_specs import RayIntervals, RaySamples


def _philox_seed_offset(device: torch.device, increment: int = 4) -> Tuple[int, int]:
    """(seed, offset) of the device's default generator, advancing it by ``increment`` -- what
    ``gen->philox_cuda_state(4)`` does in the reference (cuda/csrc/pdf.cu:376-383)."""
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    seed = int(gen.initial_seed())
    offset = int(gen.get_offset())
    gen.set_offset(offset + increment)
    return seed & 0xFFFFFFFFFFFFFFFF, offset


@torch.no_grad()
def searchsorted(
    sorted_sequence: Union[RayIntervals, RaySamples],
    values: Union[RayIntervals, RaySamples],
) -> Tuple[Tensor, Tensor]:
    """``ids_left, ids_right`` with ``seq[ids_left] <= values < seq[ids_right]`` per ray; values
    outside a ray's range get the ids of the clipped value (ref: pdf.py:13-62).

    >>> seq = RayIntervals(vals=tensor([0., 1., 0., 1., 2.]), packed_info=tensor([[0, 2], [2, 3]]))
    >>> val = RayIntervals(vals=tensor([0.5, 1.5, 2.5]), packed_info=tensor([[0, 1], [1, 2]]))
    >>> searchsorted(seq, val)
    (tensor([0, 3, 3]), tensor([1, 4, 4]))
    """
    q, k = values._to_spec(), sorted_sequence._to_spec()
    qv, kv = q["vals"].float().contiguous(), k["vals"].float().contiguous()
    dev = B.require_device(qv, kv)
    q_batched, k_batched = qv.dim() > 1, kv.dim() > 1
    if not q_batched:
        assert q["packed_info"] is not None or q["ray_indices"] is not None, \
            "flattened `values` need packed_info or ray_indices"
    if not k_batched:
        assert k["packed_info"] is not None, "flattened `sorted_sequence` needs packed_info"
    ids_left = torch.empty(qv.shape, dtype=torch.int64, device=dev)
    ids_right = torch.empty(qv.shape, dtype=torch.int64, device=dev)
    q_pi = None if q_batched else (q["packed_info"] if q["packed_info"] is not None else
                                   torch.zeros((0, 2), dtype=torch.int64, device=dev))
    with torch.cuda.device(dev):
        B.call("nfa_searchsorted", B.ptr(qv), B.ptr(q_pi), None if q_batched else B.ptr(q["ray_indices"]),
               0 if q_pi is None else q_pi.shape[0], qv.shape[-1] if q_batched else 0, qv.numel(), B.ptr(kv),
               None if k_batched else B.ptr(k["packed_info"]), kv.shape[-1] if k_batched else 0,
               B.ptr(ids_left), B.ptr(ids_right), B.stream())
    return ids_left, ids_right


@torch.no_grad()
def importance_sampling(
    intervals: RayIntervals,
    cdfs: Tensor,
    n_intervals_per_ray: Union[Tensor, int],
    stratified: bool = False,
    transform: Optional[Tuple[str, float, float]] = None,
    need_samples: bool = True,
) -> Tuple[RayIntervals, RaySamples]:
    """Inverse-transform resampling of each ray to ``n_intervals_per_ray`` intervals
    (ref: pdf.py:65-131; kernels cuda/csrc/pdf.cu:98-241).

    ``transform=(type, t_min, t_max)`` (an extension used by ``PropNetEstimator.sampling``) additionally maps the
    new edges from s in [0,1] to metric distance in the same kernel and returns
    ``(intervals, samples, t_starts, t_ends)`` with contiguous ``(n_rays, n)`` rows; values are those of the
    reference's tensor expression (estimators/prop_net.py:215-229), operation for operation.  ``need_samples=False``
    skips the sample centres (``None`` is returned in their place).

    With an int count the outputs are batched: ``intervals.vals`` (n_rays, n+1) and ``samples.vals`` (n_rays, n);
    ``n == 1`` (out of bounds upstream, pdf.cu:211) yields the ray's whole range as its single interval.  With a per-ray
    Tensor count the outputs are PACKED as the reference documents (pdf.py:92-105): ``samples.vals`` (all_samples,) with
    ``packed_info`` / ``ray_indices``, ``intervals.vals`` (all_edges,) with ``packed_info`` / ``ray_indices`` /
    ``is_left`` / ``is_right``; ray r gets ``n[r]`` samples and ``n[r] + 1`` edges (none for ``n[r] == 0``).  (The
    reference's own Tensor overload allocates zero samples, pdf.cu:324, and never worked; the semantics here are what its
    kernels compute once the allocation is right.)

    >>> iv = RayIntervals(vals=tensor([0., 1., 0., 1., 2.]), packed_info=tensor([[0, 2], [2, 3]]))
    >>> out_iv, out_sm = importance_sampling(iv, tensor([0., .5, 0., .5, 1.]), 2)
    >>> out_iv.vals, out_sm.vals
    (tensor([[0., .5, 1.], [0., 1., 2.]]), tensor([[.25, .75], [.5, 1.5]]))
    """
    if isinstance(n_intervals_per_ray, Tensor):
        assert transform is None, "the fused s -> t mapping is for batched outputs"
        return _importance_sampling_packed(intervals, cdfs, n_intervals_per_ray, stratified)
    S = int(n_intervals_per_ray)
    if S < 1:
        raise ValueError("n_intervals_per_ray must be >= 1")
    spec = intervals._to_spec()
    vals = spec["vals"].float().contiguous()
    cdfs = cdfs.float().contiguous()
    assert cdfs.numel() == vals.numel()  # pdf.cu:368
    dev = B.require_device(vals, cdfs)
    if vals.dim() > 1:
        lead = vals.shape[:-1]
        n_rays, per, pi = int(torch.Size(lead).numel()), vals.shape[-1], None
    else:
        pi = spec["packed_info"]
        assert pi is not None, "flattened intervals need packed_info"
        n_rays, per, lead = pi.shape[0], 0, (pi.shape[0],)
    out_iv = torch.empty((*lead, S + 1), dtype=torch.float32, device=dev)
    out_sm = torch.empty((*lead, S), dtype=torch.float32, device=dev) if need_samples else None
    seed, offset = _philox_seed_offset(dev) if stratified else (0, 0)
    if transform is None:
        with torch.cuda.device(dev):
            B.call("nfa_importance_sampling", B.ptr(vals), B.ptr(cdfs), B.ptr(pi), n_rays, per, S,
                   int(bool(stratified)), seed, offset, B.ptr(out_iv), B.ptr(out_sm), B.stream())
        return RayIntervals(vals=out_iv), (RaySamples(vals=out_sm) if need_samples else None)
    kind, t_min, t_max = transform
    if kind == "uniform":
        code, t_a, t_b = 1, float(t_min), float(t_max)
    elif kind == "lindisp":
        code, t_a, t_b = 2, 1 / float(t_min), 1 / float(t_max)  # reciprocals in double, rounded once, as Python does
    else:
        raise ValueError(f"Unknown transform_type: {kind}")
    t_starts = torch.empty((*lead, S), dtype=torch.float32, device=dev)
    t_ends = torch.empty_like(t_starts)
    with torch.cuda.device(dev):
        B.call("nfa_importance_sampling_t", B.ptr(vals), B.ptr(cdfs), B.ptr(pi), n_rays, per, S,
               int(bool(stratified)), seed, offset, B.ptr(out_iv), B.ptr(out_sm), code, t_a, t_b, B.ptr(t_starts),
               B.ptr(t_ends), B.stream())
    return RayIntervals(vals=out_iv), (RaySamples(vals=out_sm) if need_samples else None), t_starts, t_ends


def _importance_sampling_packed(intervals: RayIntervals, cdfs: Tensor, counts: Tensor, stratified: bool):
    """Per-ray counts -> packed outputs (ref: cuda/csrc/pdf.cu:294-355 as intended; see :func:`importance_sampling`)."""
    from .grid import _cumsum_packed
    spec = intervals._to_spec()
    vals = spec["vals"].float().contiguous()
    cdfs = cdfs.float().contiguous()
    assert cdfs.numel() == vals.numel()  # pdf.cu:305
    dev = B.require_device(vals, cdfs, counts)
    if vals.dim() > 1:
        n_rays, per, pi = int(torch.Size(vals.shape[:-1]).numel()), vals.shape[-1], None
    else:
        pi = spec["packed_info"]
        assert pi is not None, "flattened intervals need packed_info"
        n_rays, per = pi.shape[0], 0
    counts = counts.reshape(-1).to(torch.int64).contiguous()
    assert counts.shape[0] == n_rays, "n_intervals_per_ray must have one entry per ray"
    # validated BEFORE anything is sized from them: no negative count, and no samples asked of a ray without input
    # intervals (the kernel has nothing to resample there and would leave the ray's output range unwritten)
    if n_rays:
        empty_in = (pi[:, 1] < 2) if pi is not None else torch.full((n_rays,), per < 2, dtype=torch.bool, device=dev)
        bad = torch.stack([(counts < 0).any(), ((counts > 0) & empty_in).any()]).tolist()
        assert not bad[0], "negative sample count"
        assert not bad[1], "n_intervals_per_ray asks for samples on a ray that has no input interval"
    with torch.cuda.device(dev):
        totals = torch.empty(2, dtype=torch.int64, device=dev)
        iv_counts = (counts + 1) * (counts > 0)                     # pdf.cu:341-342
        sm_packed = _cumsum_packed(counts, totals[0:1])
        iv_packed = _cumsum_packed(iv_counts, totals[1:2])
        n_sm, n_iv = (int(v) for v in totals.tolist())             # the one device->host read (data_spec.hpp:91)
        sm_vals = torch.empty(n_sm, dtype=torch.float32, device=dev)
        sm_ri = torch.empty(n_sm, dtype=torch.int64, device=dev)
        iv_vals = torch.empty(n_iv, dtype=torch.float32, device=dev)
        iv_ri = torch.empty(n_iv, dtype=torch.int64, device=dev)
        iv_l = torch.empty(n_iv, dtype=torch.bool, device=dev)
        iv_r = torch.empty(n_iv, dtype=torch.bool, device=dev)
        seed, offset = _philox_seed_offset(dev) if stratified else (0, 0)
        if n_sm > 0:
            B.call("nfa_importance_sampling_packed", B.ptr(vals), B.ptr(cdfs), B.ptr(pi), n_rays, per, B.ptr(sm_packed),
                   B.ptr(iv_packed), int(bool(stratified)), seed, offset, B.ptr(sm_vals), B.ptr(sm_ri), B.ptr(iv_vals),
                   B.ptr(iv_ri), B.ptr(iv_l), B.ptr(iv_r), B.stream())
    return (RayIntervals(vals=iv_vals, packed_info=iv_packed, ray_indices=iv_ri, is_left=iv_l, is_right=iv_r),
            RaySamples(vals=sm_vals, packed_info=sm_packed, ray_indices=sm_ri))
