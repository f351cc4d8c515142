"""Inclusive / exclusive sum / product over batched or flattened (packed) tensors.

Mirrors ``nerfacc/scan.py`` of the reference (public functions :12-186, autograd Functions
:189-288): same names, arguments, asserts and gradients.  Batched inputs take the same pure
torch route as the reference; packed inputs run the flat segmented-scan engine
(csrc/segscan.hip) instead of the reference's 16-thread-per-ray Blelloch kernel.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor
from torch.autograd.function import once_differentiable

from . import _backend as B
from ._segments import SegInfo, batched_native, seginfo_from_packed

_KIND = {"inclusive_sum": 0, "exclusive_sum": 1, "inclusive_prod": 2, "exclusive_prod": 3}


def _packed_scan_raw(kind: int, reverse: bool, seg: SegInfo, inputs: Tensor, normalize: bool = False) -> Tensor:
    """One native launch; ``reverse`` = the reference's reverse-iterator launch (scan.cu:41-51)."""
    dev = B.require_device(inputs, seg.packed_info)
    inputs = inputs.contiguous()
    if inputs.dtype != torch.float32:
        raise TypeError("nerfacc_amd: packed scans support float32 only (as the reference does)")
    out = torch.empty_like(inputs)
    if inputs.numel() == 0:  # scan.cu:32-34
        return out
    with torch.cuda.device(dev):
        if seg.contiguous and not normalize:
            B.call("nfa_packed_scan", kind, int(reverse), B.ptr(seg.packed_info), B.ptr(seg.tiles), seg.n_tiles, seg.n_rays,
                   inputs.numel(), B.ptr(inputs), B.ptr(out), B.stream())
        else:
            B.call("nfa_packed_scan_generic", kind, int(reverse), int(normalize), B.ptr(seg.packed_info),
                   seg.n_rays, inputs.numel(), B.ptr(inputs), B.ptr(out), B.stream())
    return out


class _PackedSum(torch.autograd.Function):
    """ref: scan.py:189-242 (_InclusiveSum / _ExclusiveSum)."""

    @staticmethod
    def forward(ctx, inputs, seg: SegInfo, kind: int, normalize: bool):
        ctx.seg, ctx.kind, ctx.normalize = seg, kind, normalize
        return _packed_scan_raw(kind, False, seg, inputs, normalize)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_outputs):
        assert ctx.normalize is False, "Only support backward for normalize==False."  # scan.py:209
        return _packed_scan_raw(ctx.kind, True, ctx.seg, grad_outputs.contiguous()), None, None, None


class _PackedProd(torch.autograd.Function):
    """ref: scan.py:245-288 (_InclusiveProd / _ExclusiveProd); backward = scan.cu:169-214, 259-304."""

    @staticmethod
    def forward(ctx, inputs, seg: SegInfo, kind: int):
        inputs = inputs.contiguous()
        outputs = _packed_scan_raw(kind, False, seg, inputs)
        ctx.seg, ctx.kind = seg, kind
        ctx.save_for_backward(inputs, outputs)
        return outputs

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_outputs):
        inputs, outputs = ctx.saved_tensors
        seg, kind = ctx.seg, ctx.kind
        grad_outputs = grad_outputs.contiguous()
        if grad_outputs.numel() == 0:
            return torch.empty_like(grad_outputs), None, None
        if seg.contiguous:
            grad_inputs = torch.empty_like(grad_outputs)
            with torch.cuda.device(grad_outputs.device):
                B.call("nfa_packed_prod_backward", kind, B.ptr(seg.packed_info), B.ptr(seg.tiles), seg.n_tiles, seg.n_rays,
                       inputs.numel(), B.ptr(inputs), B.ptr(outputs), B.ptr(grad_outputs), B.ptr(grad_inputs),
                       B.stream())
        else:  # the reference's composition, on the generic kernel
            sum_kind = 0 if kind == 2 else 1
            grad_inputs = _packed_scan_raw(sum_kind, True, seg, grad_outputs * outputs) / inputs.clamp_min(1e-10)
        return grad_inputs, None, None


def _check_packed(inputs: Tensor, packed_info: Tensor) -> SegInfo:
    assert inputs.dim() == 1, "inputs must be flattened."
    assert packed_info.dim() == 2 and packed_info.shape[-1] == 2, "packed_info must be 2-D with shape (B, 2)."
    return seginfo_from_packed(packed_info, inputs.numel())


def inclusive_sum(inputs: Tensor, packed_info: Optional[Tensor] = None, normalize: bool = False) -> Tensor:
    """Inclusive sum along the last dim, or per chunk of a flattened tensor (ref: scan.py:12-53).

    >>> inclusive_sum(tensor([1.,2.,3.,4.,5.,6.,7.,8.,9.]), tensor([[0,2],[2,3],[5,4]]))
    tensor([ 1.,  3.,  3.,  7., 12.,  6., 13., 21., 30.])
    """
    if packed_info is None:
        seg = batched_native(inputs)
        if seg is None:
            return torch.cumsum(inputs, dim=-1)
        return _PackedSum.apply(inputs.contiguous().view(-1), seg, _KIND["inclusive_sum"], normalize).view(inputs.shape)
    seg = _check_packed(inputs, packed_info)
    return _PackedSum.apply(inputs, seg, _KIND["inclusive_sum"], normalize)


def exclusive_sum(inputs: Tensor, packed_info: Optional[Tensor] = None, normalize: bool = False) -> Tensor:
    """Exclusive sum (ref: scan.py:56-102)."""
    if packed_info is None:
        seg = batched_native(inputs)
        if seg is None:
            return torch.cumsum(torch.cat([torch.zeros_like(inputs[..., :1]), inputs[..., :-1]], dim=-1), dim=-1)
        return _PackedSum.apply(inputs.contiguous().view(-1), seg, _KIND["exclusive_sum"], normalize).view(inputs.shape)
    seg = _check_packed(inputs, packed_info)
    return _PackedSum.apply(inputs, seg, _KIND["exclusive_sum"], normalize)


def inclusive_prod(inputs: Tensor, packed_info: Optional[Tensor] = None) -> Tensor:
    """Inclusive product (ref: scan.py:105-146)."""
    if packed_info is None:
        seg = batched_native(inputs)
        if seg is None:
            return torch.cumprod(inputs, dim=-1)
        return _PackedProd.apply(inputs.contiguous().view(-1), seg, _KIND["inclusive_prod"]).view(inputs.shape)
    seg = _check_packed(inputs, packed_info)
    return _PackedProd.apply(inputs, seg, _KIND["inclusive_prod"])


def exclusive_prod(inputs: Tensor, packed_info: Optional[Tensor] = None) -> Tensor:
    """Exclusive product (ref: scan.py:149-186)."""
    if packed_info is None:
        seg = batched_native(inputs)
        if seg is None:
            return torch.cumprod(torch.cat([torch.ones_like(inputs[..., :1]), inputs[..., :-1]], dim=-1), dim=-1)
        return _PackedProd.apply(inputs.contiguous().view(-1), seg, _KIND["exclusive_prod"]).view(inputs.shape)
    seg = _check_packed(inputs, packed_info)
    return _PackedProd.apply(inputs, seg, _KIND["exclusive_prod"])
