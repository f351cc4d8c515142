"""ctypes binding of libnerfacc_hip.so (C ABI: include/nerfacc_hip.h).

This is the counterpart of the reference's ``nerfacc/cuda/__init__.py`` +
``nerfacc/cuda/_backend.py`` (lazy ``getattr(_C, name)`` over a pybind11 module):
the library is loaded on first use and every native call goes through
:func:`call`.  There is NO fallback: if the library cannot be loaded or the
tensors are not on a ROCm device the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import torch

from . import _build

_lock = threading.Lock()
_lib: Optional[C.CDLL] = None

_vp, _i64, _i32, _f32, _u64, _int = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_uint64, C.c_int


class TraverseArgs(C.Structure):
    """struct nfa_traverse_args (include/nerfacc_hip.h)."""
    _fields_ = [
        ("n_rays", _i64), ("rays_o", _vp), ("rays_d", _vp), ("rays_mask", _vp),
        ("n_grids", _i32), ("res", _i32 * 3), ("binaries", _vp), ("aabbs", _vp),
        ("hits", _vp), ("t_sorted", _vp), ("t_indices", _vp),
        ("near_planes", _vp), ("far_planes", _vp),
        ("step_size", _f32), ("cone_angle", _f32), ("traverse_steps_limit", _i32), ("mode", _i32),
        ("iv_vals", _vp), ("iv_ray_indices", _vp), ("iv_is_left", _vp), ("iv_is_right", _vp),
        ("iv_starts", _vp), ("iv_cnts", _vp),
        ("sm_vals", _vp), ("sm_ray_indices", _vp), ("sm_is_valid", _vp),
        ("sm_t_starts", _vp), ("sm_t_ends", _vp), ("sm_starts", _vp), ("sm_cnts", _vp),
        ("terminate_planes", _vp),
        ("ray_filter", _vp), ("ray_filter_min", _i32),
        ("bricks", _vp), ("coarse", _vp),
        ("steps_limit_dev", _vp), ("n_listed_dev", _vp), ("run_if_nonzero", _vp),
    ]


# name -> argtypes (all return int unless listed in _RESTYPES)
_SIGS = {
    "nfa_exclusive_cumsum_i64": [_vp, _i64, _vp, _vp, _vp, _vp],
    "nfa_exclusive_cumsum_pairs_i64": [_vp, _i64, _vp, _vp, _vp, _vp],
    "nfa_exclusive_cumsum_pairs_stats_i64": [_vp, _i64, _vp, _vp, _vp, _vp],
    "nfa_pack_info": [_vp, _i64, _i64, _vp, _vp, _vp, _vp],
    "nfa_pdf_loss_fwd": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp, _vp, _vp],
    "nfa_pdf_loss_bwd": [_vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp, _vp, _vp, _vp],
    "nfa_pdf_loss_partials": [_i64, _i32, _i32],
    "nfa_pdf_loss_sum_fwd": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp, _vp, _vp],
    "nfa_pdf_loss_mean_bwd": [_vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp, _vp, _vp, _vp],
    "nfa_pack_bits": [_vp, _i64, _vp, _vp],
    "nfa_ray_aabb_intersect": [_vp, _vp, _i64, _vp, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp],
    "nfa_ray_events": [_vp, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp],
    "nfa_traverse_grids": [C.POINTER(TraverseArgs), _vp],
    "nfa_bricks_words": [_i32, C.POINTER(_i32)],
    "nfa_pack_bricks": [_vp, _i32, C.POINTER(_i32), _vp, _vp, _vp],
    "nfa_walk_bits_words": [_i32, C.POINTER(_i32)],
    "nfa_pack_walk_bits": [_vp, _i32, C.POINTER(_i32), _vp, _vp],
    "nfa_traverse_runs": [C.POINTER(TraverseArgs), _vp, _vp, _vp, _i32, _vp, _f32, _vp, _i64, _vp],
    "nfa_bin_rays": [_vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "nfa_grid_cell_points": [_vp, _vp, _i64, C.POINTER(_i32), _vp, _vp, _vp],
    "nfa_grid_ema_update": [_vp, _i64, _vp, _i64, _vp, _f32, _vp, _vp],
    "nfa_grid_rebinarize_scratch_bytes": [],
    "nfa_grid_rebinarize": [_vp, _i32, C.POINTER(_i32), _f32, _vp, _vp, _vp, _vp],
    "nfa_expand_runs": [_i64, _f32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "nfa_fill_ray_indices": [_i64, _vp, _vp, _vp],
    "nfa_traverse_cone_runs": [C.POINTER(TraverseArgs), _vp, _vp, _i32, _vp, _vp, _i64, _vp],
    "nfa_alive_rays": [_vp, _vp, _i64, _f32, _i64, _vp, _vp, _vp, _vp],
    "nfa_testmode_begin": [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _vp],
    "nfa_testmode_alive": [_vp, _vp, _vp, _f32, _i64, _vp, _vp, _vp, _i32, _vp],
    "nfa_traverse_cone_walk": [C.POINTER(TraverseArgs), _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i64, _vp],
    "nfa_expand_cone_arena": [_vp, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp],
    "nfa_bin_rays_levels": [_vp, _vp, _i64, _vp, _i32, C.POINTER(_i32), _f32, _vp, _vp, _vp, _vp],
    "nfa_expand_cone_runs": [_i64, _f32, _f32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    "nfa_expand_intervals": [_i64, _f32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "nfa_seg_plan": [_i64, _i64, C.POINTER(_i64), C.POINTER(_i64)],
    "nfa_seg_table_rows": [_i64],
    "nfa_seg_build_tiles": [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp],
    "nfa_packed_scan": [_int, _int, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp],
    "nfa_packed_scan_generic": [_int, _int, _int, _vp, _i64, _i64, _vp, _vp, _vp],
    "nfa_packed_prod_backward": [_int, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp],
    "nfa_render_from_density_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp],
    "nfa_render_from_alpha_fwd": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp],
    "nfa_render_from_density_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp],
    "nfa_density_cdf_rows_fwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp],
    "nfa_density_cdf_rows_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp],
    "nfa_render_from_alpha_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp],
    "nfa_render_visibility": [_vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp],
    "nfa_compact_samples": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp],
    "nfa_accumulate_along_rays": [_vp, _vp, _i32, _vp, _vp, _i64, _i64, _i64, _int, _vp, _vp],
    "nfa_accumulate_along_rays_atomic": [_vp, _vp, _i32, _vp, _i64, _i64, _vp, _vp],
    "nfa_accumulate_along_rays_bwd": [_vp, _vp, _i32, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp],
    "nfa_render_accumulate_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp],
    "nfa_render_accumulate_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp],
    "nfa_render_fused_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nfa_render_fused_bwd": [_vp] * 13 + [_i64, _i64, _i64, _vp, _vp, _vp],
    "nfa_render_step_accumulate": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f32, _vp, _vp, _vp, _vp, _vp],
    "nfa_importance_sampling": [_vp, _vp, _vp, _i64, _i64, _i64, _int, _u64, _u64, _vp, _vp, _vp],
    "nfa_importance_sampling_t": [_vp, _vp, _vp, _i64, _i64, _i64, _int, _u64, _u64, _vp, _vp, _int, _f32, _f32, _vp, _vp, _vp],
    "nfa_importance_sampling_packed": [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nfa_searchsorted": [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp],
    "nfa_cumsum_scratch_bytes": [_i64],
    "nfa_last_error": [],
    "nfa_version": [],
    "nfa_set_tuning": [C.c_char_p, C.c_char_p],
    "nfa_device_arch": [C.c_char_p, _int],
}
_RESTYPES = {"nfa_grid_rebinarize_scratch_bytes": _i64, "nfa_bricks_words": _i64, "nfa_walk_bits_words": _i64, "nfa_pdf_loss_partials": _i64, "nfa_cumsum_scratch_bytes": _i64, "nfa_seg_table_rows": _i64, "nfa_seg_plan": None, "nfa_last_error": C.c_char_p}

EXPORTED_SYMBOLS = tuple(_SIGS)


def library_path() -> str:
    return _build.LIB_PATH


def load() -> C.CDLL:
    """Load (building first if the in-tree library is missing or stale and hipcc exists)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = _build.LIB_PATH
        if _build.is_stale() and _build.hipcc() is not None:
            path = _build.build()
        if not os.path.exists(path):
            raise RuntimeError(
                "nerfacc_amd: libnerfacc_hip.so is missing and hipcc is not available to build it. "
                "There is no CPU fallback for the native ops.")
        lib = C.CDLL(path)
        for name, argtypes in _SIGS.items():
            fn = getattr(lib, name)  # AttributeError => header and library disagree
            fn.argtypes = argtypes
            fn.restype = _RESTYPES[name] if name in _RESTYPES else _int
        if lib.nfa_version() != ABI_VERSION:
            raise RuntimeError(f"nerfacc_amd: {path} was built from include/nerfacc_hip.h version {lib.nfa_version()}, these "
                               f"bindings are written for {ABI_VERSION}: rebuild the library (nerfacc_amd._build.build(force=True))")
        _lib = lib
    return _lib


ABI_VERSION = 401   # include/nerfacc_hip.h: NFA_VERSION


def set_tuning(name: str, value: Optional[str]) -> None:
    """A/B knobs of the tests and measurement scripts (include/nerfacc_hip.h: nfa_set_tuning); None unsets."""
    call("nfa_set_tuning", name.encode(), None if value is None else str(value).encode())


class NativeError(RuntimeError):
    """Counterpart of the RuntimeError TORCH_CHECK raises in the reference."""


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.nfa_last_error()
        raise NativeError(f"{name}: {msg.decode() if msg else rc}")


def call_group(name: str, fn) -> None:
    """Several native calls that form ONE logical op across streams (``fn`` issues them with :func:`call`); a hook for
    instrumentation (bench.py brackets the group with one event pair on the calling stream)."""
    fn()


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_CUR_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream() -> int:
    """The current stream's handle.  Through the Stream object (``torch.cuda.current_stream().cuda_stream``) this cost 8 us per
    native call -- 40 us of every traversal, more than its kernels take in a test-mode iteration on a small scene."""
    if _RAW_STREAM is not None and _CUR_DEVICE is not None:
        return _RAW_STREAM(_CUR_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def require_device(*tensors: Optional[torch.Tensor]) -> torch.device:
    """All tensors on one ROCm device (the reference's CHECK_CUDA, utils_cuda.cuh:13-18)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise NotImplementedError(
                "nerfacc_amd: this op runs only on a ROCm device (the reference has no CPU "
                "implementation of its native ops either); got a tensor on " + str(t.device))
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"nerfacc_amd: tensors on different devices ({dev} vs {t.device})")
    if dev is None:
        raise RuntimeError("nerfacc_amd: no tensor argument")
    return dev


def seg_plan(n_elems: int, n_rays: int = 0):
    """(tile_elems, n_tiles) chosen by the library for n_rays rays with n_elems samples in all."""
    t, n = _i64(0), _i64(0)
    load().nfa_seg_plan(n_elems, n_rays, C.byref(t), C.byref(n))
    return int(t.value), int(n.value)


def cumsum_scratch(n: int, device) -> torch.Tensor:
    nbytes = ((max(n, 1) + 2047) // 2048 + 1) * 8
    return torch.empty(nbytes // 8, dtype=torch.int64, device=device)
