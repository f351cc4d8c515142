"""Checker helpers shared by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg (test infrastructure,
like everything under oracle/: never imported by the product package).

The sampler's output is compared with the oracle's restatement of nerfacc/estimators/occ_grid.py:85-221 bit for bit.
Visibility is a threshold on fp32 transmittance / opacity whose last ulp depends on the exp() implementation, so a
sample whose value lies within a guard band of a threshold may legitimately fall on either side; such samples are
identified on the oracle's side and everything else must agree exactly.
"""
from __future__ import annotations

import numpy as np


def _keys(ri, ts):
    """(ray, t_start) as one sortable uint64 (t_start >= 0: its bit pattern orders like its value)."""
    return (np.asarray(ri).astype(np.uint64) << np.uint64(32)) | np.ascontiguousarray(ts, np.float32).view(np.uint32).astype(np.uint64)


def compare_sampling(got, oracle_kept, oracle_full=None, trans=None, alphas=None, early_stop_eps=0.0, alpha_thre=0.0,
                     guard=1e-6):
    """got / oracle_kept: (ray_indices, t_starts, t_ends) of the product and of the oracle after visibility.
    oracle_full (ri, ts, te) + trans / alphas: the oracle's samples before visibility and their transmittance / opacity.
    Returns (ok, info): ok iff the outputs are identical, or differ only in samples inside the guard band."""
    g_ri, g_ts, g_te = (np.asarray(a) for a in got)
    o_ri, o_ts, o_te = (np.asarray(a) for a in oracle_kept)
    if g_ri.shape == o_ri.shape and np.array_equal(g_ri, o_ri) and np.array_equal(g_ts, o_ts) and np.array_equal(g_te, o_te):
        return True, dict(identical=True, n=int(g_ri.size), guarded=0)
    if oracle_full is None or trans is None:
        return False, dict(identical=False, n_got=int(g_ri.size), n_oracle=int(o_ri.size), guarded=None)
    f_ri, f_ts, f_te = (np.asarray(a) for a in oracle_full)
    near = np.abs(trans - np.float32(early_stop_eps)) < guard * max(1.0, float(early_stop_eps)) if early_stop_eps > 0 else np.zeros(trans.shape, bool)
    if alpha_thre > 0 and alphas is not None:
        near |= np.abs(alphas - np.float32(alpha_thre)) < guard
    # a sample on the band also decides the samples behind it on the same ray only through its own visibility (the mask
    # is per sample), so the band is exactly `near`
    vis = trans >= np.float32(early_stop_eps)
    if alpha_thre > 0 and alphas is not None:
        vis &= alphas >= np.float32(alpha_thre)
    must = _keys(f_ri[vis & ~near], f_ts[vis & ~near])
    may = _keys(f_ri[near], f_ts[near])
    gk = _keys(g_ri, g_ts)
    missing = np.setdiff1d(must, gk, assume_unique=False)
    extra = np.setdiff1d(np.setdiff1d(gk, must), may)
    # t_ends of the common samples
    fk = _keys(f_ri, f_ts)
    order = np.argsort(fk, kind="stable")
    pos = np.searchsorted(fk[order], gk)
    pos = np.clip(pos, 0, fk.size - 1)
    te_ok = bool(np.array_equal(f_te[order][pos], g_te)) if extra.size == 0 else False
    ok = missing.size == 0 and extra.size == 0 and te_ok
    return ok, dict(identical=False, n_got=int(g_ri.size), n_oracle=int(o_ri.size), guarded=int(near.sum()),
                    missing=int(missing.size), extra=int(extra.size), t_ends_equal=te_ok)
