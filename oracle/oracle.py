"""CPU oracle for the nerfacc hot path -- TEST INFRASTRUCTURE ONLY.

Thin numpy/ctypes front end over ``oracle/nerfacc_oracle.c`` (a plain-C
restatement of the reference's CUDA kernels) plus numpy restatements of the
reference's Python composites (volrend.py, pack.py, occ_grid.py sampling).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product package ``nerfacc_amd`` never
does: it fails loudly when its HIP library is missing.

All citations are relative to /root/reference/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "nerfacc_oracle.c")
_BUILD = os.path.join(_HERE, "_build")
# ORACLE_SANITIZE=1: a second build with AddressSanitizer + UndefinedBehaviorSanitizer (oracle/sanitize.sh runs the CPU
# tests against it; the sanitizer runtime must then be preloaded into python)
_SANITIZE = os.environ.get("ORACLE_SANITIZE", "0") != "0"
_SO = os.path.join(_BUILD, "libnerfacc_oracle_san.so" if _SANITIZE else "libnerfacc_oracle.so")

_lib = None


def build(force: bool = False) -> str:
    """gcc-compile the C restatement (no FMA contraction, OpenMP)."""
    os.makedirs(_BUILD, exist_ok=True)
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared", "-fPIC",
               "-fvisibility=hidden", "-o", _SO + ".tmp", _SRC, "-lm"]
        if _SANITIZE:
            cmd[1:2] = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
        subprocess.run(cmd, check=True)
        os.replace(_SO + ".tmp", _SO)
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_philox_uniform.restype = C.c_float
        _lib.orc_philox_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def max_threads() -> int:
    return int(lib().orc_max_threads())


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


def _u8(a):
    return np.ascontiguousarray(np.asarray(a).astype(np.uint8))


# --------------------------------------------------------------------------- grid
def ray_aabb_intersect(rays_o, rays_d, aabbs, near_plane=-np.inf, far_plane=np.inf, miss_value=np.inf):
    """nerfacc/grid.py:13-51 -> cuda/csrc/grid.cu:477-519."""
    rays_o, rays_d, aabbs = _f32(rays_o), _f32(rays_d), _f32(aabbs)
    n, m = rays_o.shape[0], aabbs.shape[0]
    t_mins = np.empty((n, m), np.float32)
    t_maxs = np.empty((n, m), np.float32)
    hits = np.empty((n, m), np.uint8)
    lib().orc_ray_aabb_intersect(_p(rays_o), _p(rays_d), C.c_int64(n), _p(aabbs), C.c_int64(m),
                                 C.c_float(near_plane), C.c_float(far_plane), C.c_float(miss_value),
                                 _p(t_mins), _p(t_maxs), _p(hits))
    return t_mins, t_maxs, hits.astype(bool)


def sort_intersections(t_mins, t_maxs):
    """grid.py:160-162: sort(cat([t_mins, t_maxs], -1)).  Stable (ties are
    unspecified in the reference)."""
    t = np.concatenate([t_mins, t_maxs], axis=-1)
    idx = np.argsort(t, axis=-1, kind="stable").astype(np.int64)
    return np.take_along_axis(t, idx, axis=-1), idx


def _traverse_pass(rays_o, rays_d, rays_mask, binaries, aabbs, hits, t_sorted, t_indices,
                   near_planes, far_planes, step_size, cone_angle, limit, first_pass, iv, sm, term):
    n_rays = rays_o.shape[0]
    res = np.asarray(binaries.shape[1:], dtype=np.int32)
    lib().orc_traverse_grids_pass(
        C.c_int64(n_rays), _p(rays_o), _p(rays_d), _p(rays_mask),
        C.c_int32(binaries.shape[0]), _p(res), _p(binaries), _p(aabbs),
        _p(hits), _p(t_sorted), _p(t_indices), _p(near_planes), _p(far_planes),
        C.c_float(step_size), C.c_float(cone_angle), C.c_int32(limit), C.c_int(first_pass),
        _p(iv.get("vals")), _p(iv.get("ray_indices")), _p(iv.get("is_left")), _p(iv.get("is_right")),
        _p(iv.get("chunk_starts")), _p(iv.get("chunk_cnts")),
        _p(sm.get("vals")), _p(sm.get("ray_indices")), _p(sm.get("is_valid")),
        _p(sm.get("chunk_starts")), _p(sm.get("chunk_cnts")),
        _p(term))


def _alloc_from_chunk(spec, masks, valid):
    """data_spec.hpp:86-96 memalloc_data_from_chunk (always zero-initialised here)."""
    cnts = spec["chunk_cnts"]
    cumsum = np.cumsum(cnts, dtype=np.int64)
    n = int(cumsum[-1]) if cnts.size else 0
    spec["chunk_starts"] = cumsum - cnts
    spec["vals"] = np.zeros(n, np.float32)
    spec["ray_indices"] = np.zeros(n, np.int64)
    if masks:
        spec["is_left"] = np.zeros(n, np.uint8)
        spec["is_right"] = np.zeros(n, np.uint8)
    if valid:
        spec["is_valid"] = np.zeros(n, np.uint8)


def traverse_grids(rays_o, rays_d, binaries, aabbs, near_planes=None, far_planes=None,
                   step_size=1e-3, cone_angle=0.0, traverse_steps_limit=None, over_allocate=False,
                   rays_mask=None, t_sorted=None, t_indices=None, hits=None):
    """nerfacc/grid.py:93-192 + host code cuda/csrc/grid.cu:320-474.

    Returns (intervals, samples, terminate_planes); intervals/samples are dicts
    with vals, ray_indices, packed_info, is_left/is_right or is_valid.
    """
    rays_o, rays_d, aabbs = _f32(rays_o), _f32(rays_d), _f32(aabbs)
    binaries = _u8(binaries)
    n_rays = rays_o.shape[0]
    near_planes = np.zeros(n_rays, np.float32) if near_planes is None else _f32(near_planes)
    far_planes = np.full(n_rays, np.inf, np.float32) if far_planes is None else _f32(far_planes)
    mask = np.ones(n_rays, np.uint8) if rays_mask is None else _u8(rays_mask)
    limit = -1 if traverse_steps_limit is None else int(traverse_steps_limit)
    if over_allocate:
        assert limit > 0
    if t_sorted is None or t_indices is None or hits is None:
        t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, aabbs)
        t_sorted, t_indices = sort_intersections(t_mins, t_maxs)
    t_sorted, t_indices, hits = _f32(t_sorted), _i64(t_indices), _u8(hits)

    iv, sm = {}, {}
    # The reference leaves terminate_planes uninitialised for rays it skips
    # (masked rays; zero-sample rays in the fill pass).  We define those
    # entries as the ray's near plane in over-allocate mode and as the count
    # pass's t_last in two-pass mode.
    term = near_planes.copy()
    args = (rays_o, rays_d)
    common = (binaries, aabbs, hits, t_sorted, t_indices, near_planes, far_planes,
              float(step_size), float(cone_angle), limit)
    if over_allocate:  # grid.cu:364-404
        iv["chunk_cnts"] = np.full(n_rays, limit * 2, np.int64) * mask
        _alloc_from_chunk(iv, True, False)
        sm["chunk_cnts"] = np.full(n_rays, limit, np.int64) * mask
        _alloc_from_chunk(sm, False, True)
        _traverse_pass(*args, mask, *common, 0, iv, sm, term)
        for s in (iv, sm):  # compute_chunk_start with the actual counts
            s["chunk_starts"] = np.cumsum(s["chunk_cnts"], dtype=np.int64) - s["chunk_cnts"]
    else:  # grid.cu:405-471 (rays_mask is ignored: nullptr)
        iv["chunk_cnts"] = np.zeros(n_rays, np.int64)
        sm["chunk_cnts"] = np.zeros(n_rays, np.int64)
        _traverse_pass(*args, None, *common, 1, iv, sm, term)
        _alloc_from_chunk(iv, True, False)
        _alloc_from_chunk(sm, False, True)
        _traverse_pass(*args, None, *common, 0, iv, sm, None)
    out_iv = dict(vals=iv["vals"], ray_indices=iv["ray_indices"],
                  packed_info=np.stack([iv["chunk_starts"], iv["chunk_cnts"]], -1),
                  is_left=iv["is_left"].astype(bool), is_right=iv["is_right"].astype(bool))
    out_sm = dict(vals=sm["vals"], ray_indices=sm["ray_indices"],
                  packed_info=np.stack([sm["chunk_starts"], sm["chunk_cnts"]], -1),
                  is_valid=sm["is_valid"].astype(bool))
    return out_iv, out_sm, term


# --------------------------------------------------------------------------- scan / pack
_KIND = {"inclusive_sum": 0, "exclusive_sum": 1, "inclusive_prod": 2, "exclusive_prod": 3}


def packed_scan(kind, inputs, packed_info, normalize=False, backward=False):
    """cuda/csrc/scan.cu:9-165,217-257 (forward launches and the reverse-iterator ones)."""
    inputs = _f32(inputs)
    packed_info = _i64(packed_info)
    starts = np.ascontiguousarray(packed_info[:, 0])
    cnts = np.ascontiguousarray(packed_info[:, 1])
    out = np.zeros_like(inputs)
    lib().orc_packed_scan(C.c_int(_KIND[kind]), _p(starts), _p(cnts), C.c_int64(starts.shape[0]),
                          _p(inputs), C.c_int64(inputs.shape[0]), C.c_int(bool(normalize)),
                          C.c_int(bool(backward)), _p(out))
    return out


def inclusive_sum(x, packed_info, normalize=False):
    return packed_scan("inclusive_sum", x, packed_info, normalize)


def exclusive_sum(x, packed_info, normalize=False):
    return packed_scan("exclusive_sum", x, packed_info, normalize)


def inclusive_prod(x, packed_info):
    return packed_scan("inclusive_prod", x, packed_info)


def exclusive_prod(x, packed_info):
    return packed_scan("exclusive_prod", x, packed_info)


def sum_backward(kind, grad_out, packed_info):
    """scan.py:205-214 / :233-242: same scan over reversed data."""
    return packed_scan(kind, grad_out, packed_info, False, True)


def prod_backward(kind, inputs, outputs, grad_out, packed_info):
    """scan.cu:169-214 / :259-304: reverse sum-scan of g*out, / clamp_min(in, 1e-10)."""
    sum_kind = "inclusive_sum" if kind == "inclusive_prod" else "exclusive_sum"
    g = packed_scan(sum_kind, _f32(grad_out) * _f32(outputs), packed_info, False, True)
    return g / np.maximum(_f32(inputs), np.float32(1e-10))


def pack_info(ray_indices, n_rays=None):
    """nerfacc/pack.py:38-46."""
    ray_indices = _i64(ray_indices)
    if n_rays is None:
        n_rays = int(ray_indices.max()) + 1
    out = np.zeros((n_rays, 2), np.int64)
    lib().orc_pack_info(_p(ray_indices), C.c_int64(ray_indices.shape[0]), C.c_int64(n_rays), _p(out))
    return out


# --------------------------------------------------------------------------- volrend (numpy fp32)
def render_transmittance_from_alpha(alphas, packed_info, prefix_trans=None):
    """volrend.py:200-206."""
    trans = exclusive_prod(np.float32(1.0) - _f32(alphas), packed_info)
    if prefix_trans is not None:
        trans = trans * _f32(prefix_trans)
    return trans


def render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info, prefix_trans=None):
    """volrend.py:256-264."""
    sigmas_dt = _f32(sigmas) * (_f32(t_ends) - _f32(t_starts))
    alphas = np.float32(1.0) - np.exp(-sigmas_dt)
    trans = np.exp(-exclusive_sum(sigmas_dt, packed_info))
    if prefix_trans is not None:
        trans = trans * _f32(prefix_trans)
    return trans.astype(np.float32), alphas.astype(np.float32)


def render_weight_from_alpha(alphas, packed_info, prefix_trans=None):
    """volrend.py:305-309."""
    trans = render_transmittance_from_alpha(alphas, packed_info, prefix_trans)
    return trans * _f32(alphas), trans


def render_weight_from_density(t_starts, t_ends, sigmas, packed_info, prefix_trans=None):
    """volrend.py:358-362."""
    trans, alphas = render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info, prefix_trans)
    return trans * alphas, trans, alphas


def render_weight_from_density_backward(t_starts, t_ends, sigmas, packed_info, g_w, g_t=None, g_a=None,
                                        prefix_trans=None):
    """Analytic gradient wrt sigmas in float64, used to
    check the fused HIP backward; cross-checked against torch autograd of the
    reference's batched path in tests."""
    ts, te, sg = (np.asarray(a, np.float64) for a in (t_starts, t_ends, sigmas))
    dt = te - ts
    x = sg * dt
    packed_info = _i64(packed_info)
    gw = np.asarray(g_w, np.float64)
    gt = np.zeros_like(x) if g_t is None else np.asarray(g_t, np.float64)
    ga = np.zeros_like(x) if g_a is None else np.asarray(g_a, np.float64)
    out = np.zeros_like(x)
    for s, n in packed_info:
        if n == 0:
            continue
        xs = x[s:s + n]
        T = np.exp(-(np.cumsum(xs) - xs))
        if prefix_trans is not None:
            T = T * np.asarray(prefix_trans, np.float64)[s:s + n]
        e = np.exp(-xs)
        w = T * (1 - e)
        q = gw[s:s + n] * w + gt[s:s + n] * T
        suffix_excl = np.cumsum(q[::-1])[::-1] - q
        out[s:s + n] = dt[s:s + n] * (gw[s:s + n] * T * e + ga[s:s + n] * e - suffix_excl)
    return out


def render_visibility_from_alpha(alphas, packed_info, early_stop_eps=1e-4, alpha_thre=0.0, prefix_trans=None):
    """volrend.py:412-418."""
    trans = render_transmittance_from_alpha(alphas, packed_info, prefix_trans)
    vis = trans >= np.float32(early_stop_eps)
    if alpha_thre > 0:
        vis = vis & (_f32(alphas) >= np.float32(alpha_thre))
    return vis


def render_visibility_from_density(t_starts, t_ends, sigmas, packed_info, early_stop_eps=1e-4,
                                   alpha_thre=0.0, prefix_trans=None):
    """volrend.py:474-480."""
    trans, alphas = render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info, prefix_trans)
    vis = trans >= np.float32(early_stop_eps)
    if alpha_thre > 0:
        vis = vis & (alphas >= np.float32(alpha_thre))
    return vis


def accumulate_along_rays(weights, values, ray_indices, n_rays):
    """volrend.py:532-547 (index_add_); summed in float64 in sample order."""
    w = np.asarray(weights, np.float64)
    src = w[:, None] if values is None else w[:, None] * np.asarray(values, np.float64)
    out = np.zeros((n_rays, src.shape[-1]), np.float64)
    np.add.at(out, _i64(ray_indices), src)
    return out.astype(np.float32)


def rendering(t_starts, t_ends, ray_indices, n_rays, rgbs, sigmas=None, alphas=None, render_bkgd=None):
    """volrend.py:14-158 with the callback's outputs passed in."""
    pi = pack_info(ray_indices, n_rays)
    if sigmas is not None:
        weights, trans, alphas_ = render_weight_from_density(t_starts, t_ends, sigmas, pi)
    else:
        weights, trans = render_weight_from_alpha(alphas, pi)
        alphas_ = _f32(alphas)
    colors = accumulate_along_rays(weights, rgbs, ray_indices, n_rays)
    opac = accumulate_along_rays(weights, None, ray_indices, n_rays)
    mids = ((_f32(t_starts) + _f32(t_ends))[:, None] / np.float32(2.0))
    depths = accumulate_along_rays(weights, mids, ray_indices, n_rays)
    depths = depths / np.maximum(opac, np.finfo(np.float32).eps)
    if render_bkgd is not None:
        colors = colors + _f32(render_bkgd) * (np.float32(1.0) - opac)
    return colors, opac, depths, dict(weights=weights, trans=trans, alphas=alphas_)


# --------------------------------------------------------------------------- pdf
def philox_uniform(seed, subsequence, offset):
    return float(lib().orc_philox_uniform(C.c_uint64(seed), C.c_uint64(subsequence), C.c_uint64(offset)))


def philox4x32_10(ctr, key):
    ctr = np.asarray(ctr, np.uint32)
    key = np.asarray(key, np.uint32)
    out = np.zeros(4, np.uint32)
    lib().orc_philox4x32_10(_p(ctr), _p(key), _p(out))
    return out


def importance_sampling(vals, cdfs, n_intervals_per_ray, stratified=False, packed_info=None,
                        seed=0, offset=0):
    """nerfacc/pdf.py:65-131, int overload cuda/csrc/pdf.cu:359-421.

    Returns (interval edges [R, S+1], sample centres [R, S])."""
    vals, cdfs = _f32(vals), _f32(cdfs)
    S = int(n_intervals_per_ray)
    assert S >= 2, "S == 1 reads out of bounds in the reference (pdf.cu:211)"
    if packed_info is None:
        lead = vals.shape[:-1]
        n_rays = int(np.prod(lead)) if lead else 1
        per = vals.shape[-1]
        starts = cnts = None
    else:
        packed_info = _i64(packed_info)
        starts = np.ascontiguousarray(packed_info[:, 0])
        cnts = np.ascontiguousarray(packed_info[:, 1])
        n_rays, per, lead = starts.shape[0], 0, (starts.shape[0],)
    out_i = np.empty((n_rays, S + 1), np.float32)
    out_s = np.empty((n_rays, S), np.float32)
    lib().orc_importance_sampling(_p(vals), _p(cdfs), _p(starts), _p(cnts), C.c_int64(n_rays),
                                  C.c_int64(per), C.c_int64(S), C.c_int(bool(stratified)),
                                  C.c_uint64(seed), C.c_uint64(offset), _p(out_i), _p(out_s))
    return out_i.reshape(*lead, S + 1), out_s.reshape(*lead, S)


def searchsorted(key_vals, query_vals, key_packed_info=None, query_packed_info=None, query_ray_indices=None):
    """nerfacc/pdf.py:13-62 -> cuda/csrc/pdf.cu:426-456.  Returns (ids_left, ids_right)."""
    k, q = _f32(key_vals), _f32(query_vals)
    ks = kc = qs = qc = qr = None
    k_per = q_per = 0
    if key_packed_info is not None:
        kp = _i64(key_packed_info)
        ks, kc = np.ascontiguousarray(kp[:, 0]), np.ascontiguousarray(kp[:, 1])
    else:
        k_per = k.shape[-1]
    q_rays = 0
    if query_packed_info is not None:
        qp = _i64(query_packed_info)
        qs, qc = np.ascontiguousarray(qp[:, 0]), np.ascontiguousarray(qp[:, 1])
        q_rays = qs.shape[0]
        if query_ray_indices is not None:
            qr = _i64(query_ray_indices)
    else:
        q_per = q.shape[-1]
    l = np.empty(q.shape, np.int64)
    r = np.empty(q.shape, np.int64)
    lib().orc_searchsorted(_p(q), _p(qs), _p(qc), _p(qr), C.c_int64(q_rays), C.c_int64(q_per),
                           C.c_int64(q.size), _p(k), _p(ks), _p(kc), C.c_int64(k_per), _p(l), _p(r))
    return l, r


# --------------------------------------------------------------------------- estimator sampling
def occgrid_sampling(rays_o, rays_d, binaries, aabbs, sigma_fn=None, alpha_fn=None, near_plane=0.0,
                     far_plane=1e10, t_min=None, t_max=None, render_step_size=1e-3, early_stop_eps=1e-4,
                     alpha_thre=0.0, cone_angle=0.0, occs_mean=None, return_all=False):
    """nerfacc/estimators/occ_grid.py:85-221 (stratified=False)."""
    rays_o = _f32(rays_o)
    n = rays_o.shape[0]
    near = np.full(n, near_plane, np.float32)
    far = np.full(n, far_plane, np.float32)
    if t_min is not None:
        near = np.maximum(near, _f32(t_min))
    if t_max is not None:
        far = np.minimum(far, _f32(t_max))
    iv, sm, _ = traverse_grids(rays_o, rays_d, binaries, aabbs, near, far, render_step_size, cone_angle)
    t_starts = iv["vals"][iv["is_left"]]
    t_ends = iv["vals"][iv["is_right"]]
    ray_indices = sm["ray_indices"]
    packed_info = sm["packed_info"]
    full = (ray_indices, t_starts, t_ends, packed_info)
    if (alpha_thre > 0.0 or early_stop_eps > 0.0) and (sigma_fn is not None or alpha_fn is not None):
        if occs_mean is not None:
            alpha_thre = min(alpha_thre, occs_mean)
        if sigma_fn is not None:
            sig = sigma_fn(t_starts, t_ends, ray_indices)
            masks = render_visibility_from_density(t_starts, t_ends, sig, packed_info, early_stop_eps, alpha_thre)
        else:
            al = alpha_fn(t_starts, t_ends, ray_indices)
            masks = render_visibility_from_alpha(al, packed_info, early_stop_eps, alpha_thre)
        ray_indices, t_starts, t_ends = ray_indices[masks], t_starts[masks], t_ends[masks]
    if return_all:
        return (ray_indices, t_starts, t_ends), full
    return ray_indices, t_starts, t_ends


_STEP_BUFFERS: dict = {}


def bench_step(rays_o, rays_d, binaries, aabbs, render_step_size, sigma_scale=1.0, early_stop_eps=1e-4, near_plane=0.0,
               far_plane=1e10, cone_angle=0.0):
    """bench.py's CPU baseline: one whole step (sampling with the bench's analytic density -> rendering forward -> backward
    of ``colors.sum()``), every stage an OpenMP loop over rays (``orc_step_count / _fill / _render``: see the C file).
    Returns ``(ray_indices, t_starts, t_ends) kept, n_samples_before_compaction, colors (n, 3), g_sigma (M',)`` -- views of
    buffers the next call overwrites."""
    rays_o, rays_d, aabbs = _f32(rays_o), _f32(rays_d), _f32(aabbs)
    binaries = _u8(binaries)
    n = rays_o.shape[0]
    res = np.asarray(binaries.shape[1:], dtype=np.int32)
    G = binaries.shape[0]
    t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, aabbs)
    if G == 1 and not (t_mins > t_maxs).any():      # one grid: the stable sort of (t_min, t_max) is the identity
        t_sorted = np.concatenate([t_mins, t_maxs], axis=-1)
        t_indices = np.broadcast_to(np.arange(2, dtype=np.int64), (n, 2)).copy()
    else:
        t_sorted, t_indices = sort_intersections(t_mins, t_maxs)
    near = np.full(n, near_plane, np.float32); far = np.full(n, far_plane, np.float32)
    hits8 = _u8(hits)
    geo = (C.c_int64(n), _p(rays_o), _p(rays_d), C.c_int32(G), _p(res), _p(binaries), _p(aabbs), _p(hits8), _p(t_sorted),
           _p(t_indices), _p(near), _p(far), C.c_float(render_step_size), C.c_float(cone_angle))

    def buf(name, size, dtype):
        """output arrays are kept between calls (a timing loop would otherwise spend its time in page faults: half a
        gigabyte of fresh pages per pass, first touched by all threads at once)"""
        a = _STEP_BUFFERS.get(name)
        if a is None or a.size < size or a.dtype != dtype:
            a = _STEP_BUFFERS[name] = np.empty(max(size, 1), dtype)
        return a[:size]

    cnts = buf("cnts", n, np.int64)
    lib().orc_step_count(*geo, _p(cnts))
    starts = np.cumsum(cnts) - cnts
    M = int(starts[-1] + cnts[-1]) if n else 0
    ri, ts, te = buf("ri", M, np.int64), buf("ts", M, np.float32), buf("te", M, np.float32)
    vis, kept = buf("vis", M, np.uint8), buf("kept", n, np.int64)
    lib().orc_step_fill(*geo, _p(starts), _p(cnts), C.c_float(sigma_scale), C.c_float(early_stop_eps), _p(ri), _p(ts), _p(te),
                        _p(vis), _p(kept))
    kstarts = np.cumsum(kept) - kept
    Mk = int(kstarts[-1] + kept[-1]) if n else 0
    kri, kts, kte = buf("kri", Mk, np.int64), buf("kts", Mk, np.float32), buf("kte", Mk, np.float32)
    colors, gsig = buf("colors", 3 * n, np.float32).reshape(n, 3), buf("gsig", Mk, np.float32)
    lib().orc_step_render(C.c_int64(n), _p(starts), _p(cnts), _p(ts), _p(te), _p(vis), _p(kstarts), C.c_float(sigma_scale),
                          _p(kri), _p(kts), _p(kte), _p(colors), _p(gsig))
    return (kri, kts, kte), M, colors, gsig


# --------------------------------------------------------------------------- proposal-network sampling
def test_mode_loop(max_samples, rgb_sigma_fn, rays_o, rays_d, binaries, aabbs, near_plane=0.0, far_plane=1e10,
                   render_step_size=1e-3, render_bkgd=None, cone_angle=0.0, alpha_thre=0.0, early_stop_eps=1e-4,
                   guard=1e-5):
    """The caller harness ``render_image_with_occgrid_test`` (ref: examples/utils.py:252-425) restated
    on this oracle's ``traverse_grids(over_allocate=True, rays_mask, traverse_steps_limit)``:

        n_samples = max(min(num_rays // n_alive, 64), min_samples)            (:338; min_samples 1, or 4 with a cone :312)
        traverse the alive rays for at most n_samples steps from the previous termination planes     (:342-360, :407)
        weights with prefix_trans = 1 - opacity[ray_indices]                    (:370-377)
        alpha_thre mask, in-place accumulation of rgb / opacity / depth         (:379-405)
        alive = (opacity <= 1 - early_stop_eps) & (samples taken == n_samples)  (:409-414)
        final blend with the background, depth / max(opacity, eps)             (:417-422)

    ``rgb_sigma_fn(t_starts, t_ends, ray_indices) -> (rgbs (N, 3), sigmas (N,))`` in numpy.  Returns ``(rgb, opacity, depth,
    total_samples, info)``; ``info["guard_rays"]``: rays whose opacity came within ``guard`` of the early-termination
    threshold at the end of some iteration -- for those a last-ulp difference of ``exp`` decides whether the ray stays
    alive (and, through ``n_alive``, may shift everybody's schedule); a comparison is exact only when there is none."""
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    n = rays_o.shape[0]
    opacity = np.zeros((n, 1), np.float32); depth = np.zeros((n, 1), np.float32); rgb = np.zeros((n, 3), np.float32)
    mask = np.ones(n, bool)
    min_samples = 1 if cone_angle == 0 else 4
    near = np.full(n, near_plane, np.float32); far = np.full(n, far_plane, np.float32)
    t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, aabbs)
    t_sorted, t_indices = sort_intersections(t_mins, t_maxs)
    it = total = iters = 0
    guard_rays = np.zeros(n, bool)
    per_ray = np.zeros(n, np.int64)
    thre = np.float32(1 - early_stop_eps)
    while it < max_samples:
        n_alive = int(mask.sum())
        if n_alive == 0:
            break
        ns = max(min(n // n_alive, 64), min_samples)
        it += ns
        iters += 1
        iv, sm, term = traverse_grids(rays_o, rays_d, binaries, aabbs, near, far, render_step_size, cone_angle, ns, True, mask,
                                      t_sorted, t_indices, hits)
        ts, te = iv["vals"][iv["is_left"]], iv["vals"][iv["is_right"]]
        ri = sm["ray_indices"][sm["is_valid"]]
        pi = sm["packed_info"]
        if len(ri):
            rgbs, sig = rgb_sigma_fn(ts, te, ri)
            w, _, al = render_weight_from_density(ts, te, sig, pack_info(ri, n), prefix_trans=1 - opacity[ri, 0])
            if alpha_thre > 0:
                v = al >= alpha_thre
                ri, rgbs, w, ts, te = ri[v], rgbs[v], w[v], ts[v], te[v]
            np.add.at(rgb, ri, (w[:, None] * rgbs).astype(np.float32))
            np.add.at(opacity, ri, w[:, None].astype(np.float32))
            np.add.at(depth, ri, (w[:, None] * ((ts + te)[:, None] / np.float32(2.0))).astype(np.float32))
            np.add.at(per_ray, ri, 1)
        near = term
        guard_rays |= mask & (np.abs(opacity[:, 0] - thre) <= guard)
        mask = (opacity[:, 0] <= thre) & (pi[:, 1] == ns)
        total += len(ri)
    if render_bkgd is not None:
        rgb = rgb + _f32(render_bkgd) * (1.0 - opacity)
    depth = depth / np.maximum(opacity, np.finfo(np.float32).eps)
    return rgb, opacity, depth, total, dict(guard_rays=guard_rays, samples_per_ray=per_ray, iterations=iters)


def transform_stot(transform_type, s_vals, t_min, t_max):
    """nerfacc/estimators/prop_net.py:215-229, same fp32 operation order."""
    s = _f32(s_vals)
    one = np.float32(1.0)
    t_min, t_max = np.float32(t_min), np.float32(t_max)
    if transform_type == "uniform":
        return (s * t_max + (one - s) * t_min).astype(np.float32)
    if transform_type == "lindisp":
        return (one / (s * (one / t_max) + (one - s) * (one / t_min))).astype(np.float32)
    raise ValueError(f"Unknown transform_type: {transform_type}")


def batched_transmittance_from_density(t_starts, t_ends, sigmas):
    """nerfacc/volrend.py:245-264 with packed_info=None: exp(-exclusive_sum(sigma * dt)) along the last axis
    (scan.py:84-91: cumsum of the row shifted by one)."""
    sd = (_f32(sigmas) * (_f32(t_ends) - _f32(t_starts))).astype(np.float32)
    excl = np.cumsum(np.concatenate([np.zeros_like(sd[..., :1]), sd[..., :-1]], -1), axis=-1, dtype=np.float32)
    return np.exp(-excl).astype(np.float32)


def propnet_sampling(prop_sigma_fns, prop_samples, num_samples, n_rays, near_plane, far_plane,
                     sampling_type="lindisp", stratified=False, seed=0, offset=0, return_final_vals=False):
    """nerfacc/estimators/prop_net.py:38-129 (PropNetEstimator.sampling) on the oracle's importance_sampling:
    the level loop  resample -> s->t -> proposal density -> transmittance -> cdfs = 1 - cat([T, 0]),  then the final
    resampling.  Returns (t_starts, t_ends, levels) with levels = [(interval edges in s, cdfs)] per proposal network
    (what the reference caches for the loss, :115-116).  `offset` advances by 4 per importance_sampling call as the
    reference's generator does (pdf.cu:376-383)."""
    assert len(prop_sigma_fns) == len(prop_samples)
    cdfs = np.concatenate([np.zeros((n_rays, 1), np.float32), np.ones((n_rays, 1), np.float32)], -1)
    vals = cdfs.copy()
    levels = []
    for level_fn, level_samples in zip(prop_sigma_fns, prop_samples):
        vals, _ = importance_sampling(vals, cdfs, level_samples, stratified, seed=seed, offset=offset)
        offset += 4
        t_vals = transform_stot(sampling_type, vals, near_plane, far_plane)
        t_starts, t_ends = t_vals[..., :-1], t_vals[..., 1:]
        sigmas = _f32(level_fn(t_starts, t_ends))
        assert sigmas.shape == t_starts.shape
        trans = batched_transmittance_from_density(t_starts, t_ends, sigmas)
        cdfs = (np.float32(1.0) - np.concatenate([trans, np.zeros_like(trans[:, :1])], -1)).astype(np.float32)
        levels.append((vals, cdfs))
    vals, _ = importance_sampling(vals, cdfs, num_samples, stratified, seed=seed, offset=offset)
    t_vals = transform_stot(sampling_type, vals, near_plane, far_plane)
    if return_final_vals:
        return t_vals[..., :-1], t_vals[..., 1:], levels, vals
    return t_vals[..., :-1], t_vals[..., 1:], levels


def pdf_loss_batched(q_vals, q_cdfs, k_vals, k_cdfs, eps=1e-7):
    """nerfacc/estimators/prop_net.py:232-256, the batched branch: per query interval  clip(w - w_outer, 0)^2 / (w + eps)  with
    w = the query's cdf difference and w_outer = the key's cdf difference between the enclosing key edges (searchsorted).
    Returns (loss (R, Q-1), saved) -- `saved` is what the backward needs."""
    q_cdfs, k_cdfs = _f32(q_cdfs), _f32(k_cdfs)
    il, ir = searchsorted(k_vals, q_vals)
    w = (q_cdfs[..., 1:] - q_cdfs[..., :-1]).astype(np.float32)
    il, ir = il[..., :-1], ir[..., 1:]
    w_outer = (np.take_along_axis(k_cdfs, ir, -1) - np.take_along_axis(k_cdfs, il, -1)).astype(np.float32)
    d = np.maximum(w - w_outer, np.float32(0.0)).astype(np.float32)
    loss = (d * d / (w + np.float32(eps))).astype(np.float32)
    return loss, (il, ir, w, d, k_cdfs.shape, np.float32(eps))


def pdf_loss_batched_backward(g_loss, saved):
    """d loss / d k_cdfs (the query side is detached, prop_net.py:147-148): -2 d / (w + eps) flows into the right key edge,
    +2 d / (w + eps) into the left one (torch's gather backward = scatter-add)."""
    il, ir, w, d, k_shape, eps = saved
    gwo = (-np.float32(2.0) * d / (w + eps) * _f32(g_loss)).astype(np.float32)
    g = np.zeros(k_shape, np.float32)
    rows = np.broadcast_to(np.arange(k_shape[0])[:, None], il.shape)
    np.add.at(g, (rows, ir), gwo)
    np.add.at(g, (rows, il), -gwo)
    return g


def density_cdf_backward(t_starts, t_ends, sigmas, g_cdfs):
    """Backward of PropNetEstimator.sampling's level step  cdfs = 1 - cat([T, 0]),  T = exp(-exclusive_sum(sigma * dt))
    (prop_net.py:104-113, volrend.py:245-264, scan.py:233-242): g_T = -g_cdfs[:, :-1];  d T_k / d x_j = -T_k for j < k, so
    g_x_j = -sum_{k > j} g_T_k T_k  (a reverse exclusive sum);  g_sigma = g_x * dt."""
    dt = (_f32(t_ends) - _f32(t_starts)).astype(np.float32)
    T = batched_transmittance_from_density(t_starts, t_ends, sigmas)
    gT = (-_f32(g_cdfs)[..., :-1]).astype(np.float32)
    a = (gT * T).astype(np.float32)
    rev_incl = np.cumsum(a[..., ::-1], axis=-1, dtype=np.float32)[..., ::-1]
    g_x = (-(rev_incl - a)).astype(np.float32)
    return (g_x * dt).astype(np.float32)


def propnet_loss(levels, final_vals, final_trans, loss_scaler=1.0):
    """nerfacc/estimators/prop_net.py:131-154 (compute_loss): sum over the proposal levels of the MEAN of _pdf_loss(final
    intervals, cdfs of the final transmittance (detached), level intervals, level cdfs).  Returns (loss, [d loss / d level cdfs])."""
    final_trans = _f32(final_trans)
    q_cdfs = (np.float32(1.0) - np.concatenate([final_trans, np.zeros_like(final_trans[:, :1])], -1)).astype(np.float32)
    total = np.float64(0.0)
    grads = []
    for k_vals, k_cdfs in levels:
        l, saved = pdf_loss_batched(final_vals, q_cdfs, k_vals, k_cdfs)
        total += np.float64(l.astype(np.float64).mean())
        grads.append(pdf_loss_batched_backward(np.full(l.shape, np.float32(loss_scaler / l.size), np.float32), saved))
    return float(total * loss_scaler), grads


# --------------------------------------------------------------------------- occupancy-grid maintenance
def grid_cell_points(indices, jitter, res, aabb):
    """nerfacc/estimators/occ_grid.py:383-391: x = aabb_lo + (grid_coords + jitter) / resolution * (aabb_hi - aabb_lo), fp32."""
    idx = np.asarray(indices, np.int64)
    res = [int(v) for v in res]
    cz = idx % res[2]; cy = (idx // res[2]) % res[1]; cx = idx // (res[1] * res[2])
    coords = np.stack([cx, cy, cz], -1).astype(np.float32)
    aabb = _f32(aabb)
    unit = ((coords + _f32(jitter)) / np.asarray(res, np.float32)).astype(np.float32)
    return (aabb[:3] + unit * (aabb[3:] - aabb[:3])).astype(np.float32)


def grid_ema_update(occs, cell_ids, occ, ema_decay):
    """:393-398 `occs[cell_ids] = maximum(occs[cell_ids] * ema_decay, occ)`.  Cells listed more than once: the
    reference's index_put keeps an arbitrary one of the candidates; the restatement (and the product) keep the largest."""
    occs = _f32(occs).copy()
    cell_ids = np.asarray(cell_ids, np.int64)
    cand = np.maximum(occs[cell_ids] * np.float32(ema_decay), _f32(occ)).astype(np.float32)
    occs[cell_ids] = -np.inf
    np.maximum.at(occs, cell_ids, cand)
    return occs


def grid_rebinarize(occs, shape, occ_thre):
    """:403-404 thre = clamp(occs[occs >= 0].mean(), max=occ_thre); binaries = occs > thre.  Returns (binaries, thre)."""
    occs = _f32(occs)
    sel = occs[occs >= 0]
    mean = np.float32(sel.astype(np.float64).mean()) if sel.size else np.float32(np.nan)
    thre = mean if np.isnan(mean) else np.float32(min(mean, np.float32(occ_thre)))
    return (occs > thre).reshape(shape), thre


def mark_invisible_cells(occs, res, aabbs, K, c2w, width, height, near_plane=0.0):
    """:262-332 for every cell of every level (occs >= 0 initially): occs = 0 if some camera sees the cell's corner
    position x = coords / (res - 1) and no camera has it in front of its near plane, else -1 (float64 restatement of the
    projection: the comparison with the product allows the cells that sit on an image border or on the near plane)."""
    occs = _f32(occs).copy()
    res = [int(v) for v in res]
    K = np.asarray(K, np.float64); c2w = np.asarray(c2w, np.float64)
    n_cams = c2w.shape[0]
    if K.shape[0] == 1:
        K = np.repeat(K, n_cams, 0)
    R = np.transpose(c2w[:, :3, :3], (0, 2, 1))
    t = -R @ c2w[:, :3, 3:]
    cells = res[0] * res[1] * res[2]
    cx, cy, cz = np.meshgrid(np.arange(res[0]), np.arange(res[1]), np.arange(res[2]), indexing="ij")
    coords = np.stack([cx, cy, cz], -1).reshape(-1, 3).astype(np.float64)
    for lvl in range(len(aabbs)):
        ab = np.asarray(aabbs[lvl], np.float64)
        x = coords / (np.asarray(res, np.float64) - 1)
        w = (ab[:3] + x * (ab[3:] - ab[:3])).T
        uvd = K @ (R @ w + t)
        uv = uvd[:, :2] / uvd[:, 2:]
        in_img = (uvd[:, 2] >= 0) & (uv[:, 0] >= 0) & (uv[:, 0] < width) & (uv[:, 1] >= 0) & (uv[:, 1] < height)
        covered = ((uvd[:, 2] >= near_plane) & in_img).sum(0) / n_cams > 0
        too_near = ((uvd[:, 2] < near_plane) & in_img).any(0)
        keep = occs[lvl * cells:(lvl + 1) * cells] >= 0
        new = np.where(covered & ~too_near, np.float32(0.0), np.float32(-1.0))
        seg = occs[lvl * cells:(lvl + 1) * cells]
        seg[keep] = new[keep]
    return occs


# --------------------------------------------------------------------------- packed (per-ray count) resampling, packed loss
def importance_sampling_packed(vals, cdfs, counts, packed_info=None):
    """The Tensor-count overload of importance_sampling as its kernels define it (cuda/csrc/pdf.cu:98-241 with
    samples.chunk_cnts = counts; the host code at :324 allocates nothing upstream): ray r -> counts[r] samples and
    counts[r] + 1 edges, packed.  A single sample's interval is the ray's whole range (our definition; :211 reads out of
    bounds).  Returns dicts like traverse_grids' (vals, packed_info, ray_indices[, is_left, is_right])."""
    vals, cdfs = _f32(vals), _f32(cdfs)
    counts = np.asarray(counts, np.int64).reshape(-1)
    R = counts.shape[0]
    if packed_info is None:
        v2, c2 = vals.reshape(R, -1), cdfs.reshape(R, -1)
        rows = [(v2[r], c2[r]) for r in range(R)]
    else:
        pi = _i64(packed_info)
        rows = [(vals[pi[r, 0]:pi[r, 0] + pi[r, 1]], cdfs[pi[r, 0]:pi[r, 0] + pi[r, 1]]) for r in range(R)]
    sm_v, sm_r, iv_v, iv_r, iv_l, iv_rt = [], [], [], [], [], []
    for r in range(R):
        S = int(counts[r])
        v, c = rows[r]
        if S == 0 or v.size == 0:
            continue
        if S == 1:
            u = c[0] + (np.float32(0) + np.float32(0.5)) * ((c[-1] - c[0]) / np.float32(1))   # pdf.cu:133-145, n = 1
            p = int(np.searchsorted(c[:-1], u, side="right"))
            p0, p1 = min(max(p - 1, 0), v.size - 1), min(max(p, 0), v.size - 1)
            if c[p1] - c[p0] < np.float32(1e-10):
                t = (v[p0] + v[p1]) * np.float32(0.5)
            else:
                t = (u - c[p0]) * ((v[p1] - v[p0]) / (c[p1] - c[p0])) + v[p0]
            sm, iv = np.array([t], np.float32), np.array([v[0], v[-1]], np.float32)
        else:
            iv, sm = importance_sampling(v[None], c[None], S)
            iv, sm = iv[0], sm[0]
        sm_v.append(sm); sm_r.append(np.full(S, r, np.int64))
        iv_v.append(iv); iv_r.append(np.full(S + 1, r, np.int64))
        l = np.ones(S + 1, bool); l[-1] = False
        rt = np.ones(S + 1, bool); rt[0] = False
        iv_l.append(l); iv_rt.append(rt)
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    has = np.array([rows[r][0].size > 0 for r in range(R)])
    sm_c = counts * has
    iv_c = (counts + 1) * (counts > 0) * has
    return (dict(vals=cat(iv_v, np.float32), ray_indices=cat(iv_r, np.int64), is_left=cat(iv_l, bool), is_right=cat(iv_rt, bool),
                 packed_info=np.stack([np.cumsum(iv_c) - iv_c, iv_c], -1)),
            dict(vals=cat(sm_v, np.float32), ray_indices=cat(sm_r, np.int64), packed_info=np.stack([np.cumsum(sm_c) - sm_c, sm_c], -1)))


def pdf_loss_packed(q_vals, q_cdfs, q_packed_info, q_is_left, q_is_right, k_vals, k_cdfs, k_packed_info, eps=1e-7):
    """nerfacc/estimators/prop_net.py:244-256, the flattened branch: w = cdf[is_right] - cdf[is_left] of the query,
    w_outer from the key CDF at the searchsorted ids, loss = clip(w - w_outer, 0)^2 / (w + eps)."""
    il, ir = searchsorted(k_vals, q_vals, key_packed_info=k_packed_info, query_packed_info=q_packed_info)
    q_cdfs, k_cdfs = _f32(q_cdfs), _f32(k_cdfs)
    w = q_cdfs[q_is_right] - q_cdfs[q_is_left]
    w_outer = k_cdfs[ir[q_is_right]] - k_cdfs[il[q_is_left]]
    return (np.clip(w - w_outer, 0, None) ** 2 / (w + np.float32(eps))).astype(np.float32)
