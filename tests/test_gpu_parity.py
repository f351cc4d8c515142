"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs, against the committed golden fixtures (reference outputs), and -- mirroring the
reference's own tests -- against torch on equal-length chunks.

Bars (BASELINE.json north_star): bit-exact for ray_indices / packed_info / masks / sample
positions; <= 1e-5 (relative to magnitude) for fp32 weights, transmittance, colours.
"""
import os

import numpy as np
import pytest
import torch

import nerfacc_amd as na
from conftest import assert_close, load_golden, seeded_case

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ----------------------------------------------------------------------------- native lib really loaded
def test_native_library_is_loaded(dev):
    from nerfacc_amd import _backend as B
    lib = B.load()
    import ctypes as C
    buf = C.create_string_buffer(64)
    assert lib.nfa_device_arch(buf, 64) == 0
    assert buf.value.decode().startswith("gfx"), buf.value
    maps = open("/proc/self/maps").read()
    assert "libnerfacc_hip.so" in maps


# ----------------------------------------------------------------------------- cone-angle walk: its three kernels agree
def test_cone_walk_forms_agree_and_match_oracle(dev, oracle):
    """Distance-dependent steps: the count pass exists over the brick-packed grid (grid.hip: traverse_kernel<EMIT_RUNS>, one ray
    per lane; traverse_refill_kernel for limited walks) and over the 1-bit grid copy (walk.hip: cone_walk_kernel,
    cone_refill_kernel).  Every form must produce the same samples, counts and termination planes, bit for bit -- and the
    oracle's: nested levels, one level with the in-kernel slab test, masks, step limits from 1 to 40, a wide and a narrow cone,
    resolutions that are not multiples of 4, rays with zero direction components and rays that miss."""
    rng = np.random.default_rng(int(os.environ.get("NFA_CONE_SEED", "77")))   # NFA_CONE_SEED: soak runs with other seeds
    cases = [  # (levels, res, occupancy, n_rays, step, cone, limit, masked, inside)
        (3, (32, 32, 32), 0.15, 20_000, 6e-3, 0.01, None, False, True),
        (3, (32, 32, 32), 0.03, 20_000, 6e-3, 0.004, 4, True, True),
        (1, (48, 40, 30), 0.05, 9_000, 4e-3, 0.02, 1, True, False),
        (1, (64, 64, 64), 0.5, 5_000, 3e-3, 0.003, None, False, False),
        (4, (16, 16, 16), 0.02, 30_000, 1e-2, 0.004, 7, True, True),
        (2, (50, 24, 30), 0.3, 3_000, 5e-3, 0.05, 40, False, True),
        (5, (16, 16, 16), 0.05, 6_000, 1e-2, 0.004, 5, True, True),    # ten events per ray: the refill kernel reads them from memory
    ]
    from nerfacc_amd import _backend as NB
    saved = (na.grid.CONE_WALK,)
    try:
        for levels, res, occ, R, step, cone, limit, masked, inside in cases:
            est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=list(res), levels=levels).to(dev)
            b = rng.random((levels, *res)) < occ
            o = (rng.random((R, 3)).astype(np.float32) - 0.5) * (1.0 if inside else 4.0)
            d = rng.standard_normal((R, 3)).astype(np.float32)
            d[rng.random(R) < 0.05, int(rng.integers(0, 3))] = 0.0
            d /= np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-6)
            near = np.full(R, 0.05, np.float32); far = np.full(R, 1e10, np.float32)
            mask = (rng.random(R) < 0.6) if masked else None
            kw = dict(rays_mask=None if mask is None else T(mask, dev), traverse_steps_limit=limit, return_terminate=True,
                      near_hint=0.05)
            args = (T(o, dev), T(d, dev), T(b, dev), est.aabbs, T(near, dev), T(far, dev), step, cone)
            outs = {}
            for form, (walk, refill) in {"walk": (True, None), "walk, one ray per lane": (True, "0"), "bricks": (False, None),
                                         "bricks, one ray per lane": (False, "0"), "walk, small chunks": (True, "64,40")}.items():
                na.grid.CONE_WALK = walk
                NB.set_tuning("NFA_REFILL", refill)
                outs[form] = na.grid._traverse_samples(*args, **kw)
            ref = outs["walk"]
            assert ref[0].numel() > 500, (levels, res, ref[0].numel())
            for form, got in outs.items():
                assert all(torch.equal(x, y) for x, y in zip(ref, got)), (form, levels, res, limit)
            # ... and the oracle's samples
            ab = est.aabbs.cpu().numpy()
            if limit is None:
                riv, rsm, _ = oracle.traverse_grids(o, d, b, ab, near_planes=near, far_planes=far, step_size=step, cone_angle=cone)
                keep = np.ones(rsm["ray_indices"].shape[0], bool)
            else:
                riv, rsm, _ = oracle.traverse_grids(o, d, b, ab, near_planes=near, far_planes=far, step_size=step, cone_angle=cone,
                                                    traverse_steps_limit=limit, over_allocate=True,
                                                    rays_mask=mask if mask is not None else np.ones(R, bool))
                keep = rsm["is_valid"]
            L, Rr = riv["vals"][riv["is_left"]], riv["vals"][riv["is_right"]]
            assert (ref[0].cpu().numpy() == rsm["ray_indices"][keep]).all()
            assert (ref[1].cpu().numpy() == L).all() and (ref[2].cpu().numpy() == Rr).all()
            # rays without geometry (NaN / inf origin or direction): no samples in any form (DESIGN: divergences), nobody hangs
            o2, d2 = o.copy(), d.copy()
            o2[0, 0] = np.nan; o2[1, 2] = np.inf; d2[2, 1] = np.nan; d2[3, 0] = -np.inf; d2[4] = np.nan
            args2 = (T(o2, dev), T(d2, dev)) + args[2:]
            outs2 = []
            for walk, refill in ((True, None), (False, None), (True, "0")):
                na.grid.CONE_WALK = walk
                NB.set_tuning("NFA_REFILL", refill)
                outs2.append(na.grid._traverse_samples(*args2, **kw))
            assert all(torch.equal(x, y) for got in outs2[1:] for x, y in zip(outs2[0], got))
            assert (outs2[0][3][:5, 1] == 0).all()
    finally:
        na.grid.CONE_WALK = saved[0]
        NB.set_tuning("NFA_REFILL", None)


# ----------------------------------------------------------------------------- pack / scans
def test_binned_ray_assignment_same_results(dev):
    """estimator.bin_rays: the lane -> ray assignment of the walk is a permutation sorted by path length; every output
    of the sampler is exactly what it is without it (unrelated rays, rays missing the box, a mask with a step limit)."""
    import bench
    rng = np.random.default_rng(12)
    R = 50_000
    o = rng.standard_normal((R, 3)).astype(np.float32)
    d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d[:50, 1] = 0.0
    b = T(bench.make_grid(64, "shell10"), dev)
    ab = T(np.array([[-1, -1, -1, 1, 1, 1]], np.float32), dev)
    ro, rd = T(o, dev), T(d, dev)
    near, far = torch.zeros(R, device=dev), torch.full((R,), 1e10, device=dev)
    order = torch.empty(R, dtype=torch.int32, device=dev)
    from nerfacc_amd import _backend as B
    scratch = torch.empty(1024 + R, dtype=torch.uint8, device=dev)
    B.call("nfa_bin_rays", B.ptr(ro), B.ptr(rd), R, B.ptr(ab[0]), B.ptr(order), B.ptr(scratch), B.stream())
    assert torch.equal(torch.sort(order.long())[0], torch.arange(R, device=dev))          # a permutation
    tmin, tmax, hit = na.ray_aabb_intersect(ro, rd, ab)
    plen = torch.where(hit[:, 0], tmax[:, 0] - tmin[:, 0].clamp_min(0), torch.zeros(R, device=dev)).clamp_min(0)
    bins = torch.where(plen > 0, (1 + (plen / 12 ** 0.5 * 255).long()).clamp(1, 255), torch.zeros(R, dtype=torch.long, device=dev))
    bins = bins[order.long()]
    assert (bins[1:] - bins[:-1] >= -1).all() and bins[0] == 0 and bins[-1] >= 200       # sorted by bin (+-1 at bin edges)
    for kw in (dict(), dict(rays_mask=T(rng.random(R) < 0.6, dev), traverse_steps_limit=7)):
        ref = na.grid._traverse_samples(ro, rd, b, ab, near, far, 4e-3, 0.0, return_terminate=True, **kw)
        got = na.grid._traverse_samples(ro, rd, b, ab, near, far, 4e-3, 0.0, return_terminate=True, bin_rays=True, **kw)
        assert ref[0].numel() > 30_000 and all(torch.equal(x, y) for x, y in zip(ref, got))
    # distance-dependent steps through nested levels (the cone-angle walk: nfa_bin_rays_levels, key = cell boundaries crossed
    # summed over the levels): rays from inside the finest box, same results with and without the binned assignment
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=32, levels=3).to(dev)
    b3 = T(rng.random((3, 32, 32, 32)) < 0.15, dev)
    oi = T((rng.random((R, 3)).astype(np.float32) - 0.5), dev)
    nearc = torch.full((R,), 0.1, device=dev)
    ref = na.grid._traverse_samples(oi, rd, b3, est.aabbs, nearc, far, 6e-3, 0.01, near_hint=0.1)
    got = na.grid._traverse_samples(oi, rd, b3, est.aabbs, nearc, far, 6e-3, 0.01, near_hint=0.1, bin_rays=True)
    assert ref[0].numel() > 100_000 and all(torch.equal(x, y) for x, y in zip(ref, got))
    order = torch.empty(R, dtype=torch.int32, device=dev)
    import ctypes as C
    coh = torch.zeros(2, dtype=torch.int64, device=dev)
    B.call("nfa_bin_rays_levels", B.ptr(oi), B.ptr(rd), R, B.ptr(est.aabbs), 3, (C.c_int32 * 3)(32, 32, 32), 0.1, B.ptr(order),
           B.ptr(scratch), B.ptr(coh), B.stream())
    assert 1.2 < 64.0 * float(coh[0]) / float(coh[1]) < 4.0                              # unrelated rays: a wave's longest ray vs its mean
    assert torch.equal(torch.sort(order.long())[0], torch.arange(R, device=dev))          # a permutation
    # rays in walk order cross more and more cells: path length inside the finest box (res / extent = 16 boundaries per unit
    # length and axis) + outside it (8, 4 per unit), roughly
    tmin, tmax, hit = na.ray_aabb_intersect(oi, rd, est.aabbs)
    seg = (tmax - tmin.clamp_min(0.1)).clamp_min(0)
    cells = (seg[:, 0] * 16 + (seg[:, 1] - seg[:, 0]) * 8 + (seg[:, 2] - seg[:, 1]) * 4)[order.long()]
    q = R // 4
    assert cells[:q].mean() < cells[q:2 * q].mean() < cells[2 * q:3 * q].mean() < cells[3 * q:].mean()


def test_degenerate_batches(dev):
    """No rays, one ray, rays that all miss the grid and an all-empty grid through every sampler path (constant step,
    cone angle, binned assignment, mask + limit), and the new entry points with n_rays = 0."""
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=32, levels=2).to(dev)
    est.binaries = torch.rand((2, 32, 32, 32), device=dev) < 0.3
    est.bin_rays = True
    for R in (0, 1, 5000):
        ro = torch.randn(R, 3, device=dev) * 0.3
        rd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
        for cone in (0.0, 0.01):
            ri, ts, te = est.sampling(ro, rd, render_step_size=0.01, cone_angle=cone)
            assert ri.dtype == torch.int64 and ts.shape == te.shape == ri.shape and (R == 0) == (ri.numel() == 0)
            far_o = ro + 100.0                                                           # every ray misses
            ri, ts, te = est.sampling(far_o, rd.abs(), render_step_size=0.01, cone_angle=cone)
            assert ri.numel() == 0 and ts.numel() == 0
        mask = torch.zeros(R, dtype=torch.bool, device=dev)
        out = na.grid._traverse_samples(ro, rd, est.binaries, est.aabbs, torch.zeros(R, device=dev),
                                        torch.full((R,), 1e10, device=dev), 0.01, 0.0, rays_mask=mask,
                                        traverse_steps_limit=4, return_terminate=True, bin_rays=True)
        assert out[0].numel() == 0 and out[3].shape == (R, 2) and (out[3] == 0).all()
    est.binaries = torch.zeros_like(est.binaries)
    for cone in (0.0, 0.01):
        ri, ts, te = est.sampling(ro, rd, render_step_size=0.01, cone_angle=cone)
        assert ri.numel() == 0
    from nerfacc_amd import _backend as B
    z = torch.empty(0, device=dev)
    B.call("nfa_fill_ray_indices", 0, None, None, B.stream())
    B.call("nfa_bin_rays", None, None, 0, None, None, None, B.stream())
    B.call("nfa_expand_cone_runs", 0, 0.01, 0.01, None, None, 32, None, None, None, None, B.stream())


def test_automatic_ray_binning_decision(dev):
    """bin_rays = None: the estimator measures, on the counts the traversal produces anyway, how much longer a wave of 64
    neighbouring rays runs than its average ray, and bins the next batch when that exceeds 5 -- image-ordered rays stay
    unbinned, unrelated rays get binned from the second call on, and the results never change."""
    import bench
    R = 256 * 256
    b = bench.make_grid(64, "shell10")
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=64).to(dev)
    est.binaries = T(b, dev)
    assert est.bin_rays is None
    o, d = bench.make_rays(R, "image", rank=0)
    for _ in range(2):
        est.sampling(T(o, dev), T(d, dev), render_step_size=0.01)
    assert 1.0 <= est._walk_stats["max_over_mean"] < 4.0
    o, d = bench.make_rays(R, "random", rank=0)
    ro, rd = T(o, dev), T(d, dev)
    first = est.sampling(ro, rd, render_step_size=0.01)
    assert est._walk_stats["max_over_mean"] > 6.0
    second = est.sampling(ro, rd, render_step_size=0.01)                               # binned now
    est.bin_rays = False
    third = est.sampling(ro, rd, render_step_size=0.01)
    assert all(torch.equal(x, y) and torch.equal(x, z) for x, y, z in zip(first, second, third))
    # the measure itself: 64 * sum of group maxima / sum
    cnts = torch.randint(0, 50, (100_000,), device=dev)
    meta = torch.zeros(3, dtype=torch.int64, device=dev)
    na.grid._cumsum_packed(cnts, meta, stats=True)
    pad = torch.cat([cnts, cnts.new_zeros((-cnts.numel()) % 2048)]).view(-1, 2048)[::8]     # every 8th block of 2048
    assert meta.tolist() == [int(cnts.sum()), int(pad.reshape(-1, 64).max(1)[0].sum()), int(pad.sum())]


def test_long_runs_of_empty_rays(dev):
    """Blocks of hundreds to hundreds of thousands of rays without samples (finished rays of the test-mode loop,
    background pixels of an image-order batch): a tile ends after 256 rays, so such a block is spread over many tiles
    instead of being walked by one wave.  Every packed op, forward and reverse scans and the fused rendering passes with gradients, must give
    for the rays that have samples exactly what it gives on the batch without the empty rays, and zeros elsewhere."""
    g = torch.Generator(device=dev); g.manual_seed(11)
    R = 300_000
    patterns = []
    c = torch.zeros(R, dtype=torch.int64, device=dev); c[120_000:121_500] = 37; patterns.append(c)          # one block alive
    c = torch.zeros(R, dtype=torch.int64, device=dev); c[:700] = 5; c[-900:] = 9; patterns.append(c)        # huge middle gap
    c = torch.randint(1, 40, (R,), generator=g, device=dev)
    for a, b in ((10, 74), (1000, 1129), (5000, 5700), (50_000, 250_000), (299_000, 300_000)):
        c[a:b] = 0
    patterns.append(c)                                                                                       # gaps of 64 .. 200 k
    for cnts in patterns:
        has = cnts > 0
        rows = torch.nonzero(has).squeeze(1)
        ri = torch.repeat_interleave(torch.arange(R, device=dev), cnts)
        ri_c = torch.repeat_interleave(torch.arange(rows.numel(), device=dev), cnts[rows])
        n = ri.numel()
        ts = torch.rand(n, generator=g, device=dev); te = ts + 0.02
        sig = (torch.rand(n, generator=g, device=dev) * 4).requires_grad_(True)
        rgbs = torch.rand(n, 3, generator=g, device=dev).requires_grad_(True)
        sig_c, rgbs_c = sig.detach().clone().requires_grad_(True), rgbs.detach().clone().requires_grad_(True)
        full = na.rendering(ts, te, ri, n_rays=R, rgb_sigma_fn=lambda a, b, c_: (rgbs, sig))
        comp = na.rendering(ts, te, ri_c, n_rays=rows.numel(), rgb_sigma_fn=lambda a, b, c_: (rgbs_c, sig_c))
        gw = torch.rand(R, 3, generator=g, device=dev)
        ((full[0] * gw).sum() + (full[2] * gw[:, :1]).sum()).backward()
        ((comp[0] * gw[rows]).sum() + (comp[2] * gw[rows, :1]).sum()).backward()
        for k in range(3):
            assert torch.equal(full[k][rows], comp[k]) and (full[k][~has] == 0).all()
        assert torch.equal(full[3]["weights"], comp[3]["weights"]) and torch.equal(full[3]["trans"], comp[3]["trans"])
        assert torch.equal(sig.grad, sig_c.grad) and torch.equal(rgbs.grad, rgbs_c.grad)
        x = torch.rand(n, generator=g, device=dev).requires_grad_(True)
        x_c = x.detach().clone().requires_grad_(True)
        pi = na.pack_info(ri, R); pi_c = na.pack_info(ri_c, rows.numel())
        y, y_c = na.exclusive_sum(x, pi), na.exclusive_sum(x_c, pi_c)
        z, z_c = na.inclusive_prod(0.5 + 0.5 * x, pi), na.inclusive_prod(0.5 + 0.5 * x_c, pi_c)
        (y * ts + z).sum().backward(); (y_c * ts + z_c).sum().backward()                                     # reverse scans
        assert torch.equal(y, y_c) and torch.equal(z, z_c) and torch.equal(x.grad, x_c.grad)
        vis = na.render_visibility_from_density(ts, te, sig.detach(), ray_indices=ri, n_rays=R, early_stop_eps=0.05, alpha_thre=0.01)
        vis_c = na.render_visibility_from_density(ts, te, sig.detach(), ray_indices=ri_c, n_rays=rows.numel(), early_stop_eps=0.05, alpha_thre=0.01)
        assert torch.equal(vis, vis_c)
        acc = na.accumulate_along_rays(full[3]["weights"].detach(), None, ri, R)
        assert torch.equal(acc[rows], na.accumulate_along_rays(comp[3]["weights"].detach(), None, ri_c, rows.numel())) and (acc[~has] == 0).all()


def test_exclusive_cumsum_i64(dev):
    """nfa_exclusive_cumsum_i64 / _pairs_i64 vs torch.cumsum: empty, one element, block edges, the two-launch form
    (every workgroup adds up the partial sums before it) and, beyond 4 M elements, the three-launch form with a spine."""
    g = torch.Generator(device=dev); g.manual_seed(9)
    for n in (0, 1, 2047, 2048, 2049, 333_333, 1 << 20, (1 << 22) + 4097):
        cnts = torch.randint(0, 1000, (n,), generator=g, device=dev, dtype=torch.int64)
        total = torch.full((1,), -7, dtype=torch.int64, device=dev)
        starts = na.grid._exclusive_cumsum(cnts, total)
        ref = torch.cumsum(cnts, 0) - cnts
        assert torch.equal(starts, ref) and int(total) == int(cnts.sum())
        total.fill_(-7)
        packed = na.grid._cumsum_packed(cnts, total)
        assert torch.equal(packed[:, 0], ref) and torch.equal(packed[:, 1], cnts) and int(total) == int(cnts.sum())


def test_pack_info(dev, oracle):
    # tests/test_pack.py:8-18
    ri = torch.tensor([0, 2, 2, 2, 2], dtype=torch.int64, device=dev)
    assert na.pack_info(ri, n_rays=3).tolist() == [[0, 1], [1, 0], [1, 4]]
    assert na.pack_info(torch.tensor([0, 0, 1, 1, 1, 2, 2, 2, 2], device=dev), 3).tolist() == [[0, 2], [2, 3], [5, 4]]
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 200, size=5000)
    lens[rng.random(5000) < 0.3] = 0
    ri = np.repeat(np.arange(5000), lens).astype(np.int64)
    assert (na.pack_info(T(ri, dev), 5000).cpu().numpy() == oracle.pack_info(ri, 5000)).all()
    # n_rays inferred, int32 indices keep their dtype (pack.py:40-41)
    p = na.pack_info(T(ri.astype(np.int32), dev))
    assert p.dtype == torch.int32 and p.shape[0] == ri.max() + 1
    assert na.pack_info(torch.zeros(0, dtype=torch.int64, device=dev), 4).tolist() == [[0, 0]] * 4


def _equal_chunks(dev, rows=5, cols=1000):
    torch.manual_seed(42)
    data = torch.rand((rows, cols), device=dev, requires_grad=True)
    starts = torch.arange(0, data.numel(), cols, device=dev, dtype=torch.long)
    cnts = torch.full((rows,), cols, dtype=torch.long, device=dev)
    return data, torch.stack([starts, cnts], dim=-1)


@pytest.mark.parametrize("name,atol", [("inclusive_sum", 1e-8), ("exclusive_sum", 3e-4), ("inclusive_prod", 1e-8),
                                       ("exclusive_prod", 1e-8)])
def test_scans_like_reference_tests(dev, name, atol):
    # tests/test_scan.py:8-124: packed == batched torch on 5 x 1000, values and gradients
    fn = getattr(na, name)
    data, packed_info = _equal_chunks(dev)
    out1 = fn(data).flatten()
    out1.sum().backward()
    g1 = data.grad.clone()
    data.grad.zero_()
    out2 = fn(data.flatten(), packed_info=packed_info)
    out2.sum().backward()
    g2 = data.grad.clone()
    assert torch.allclose(out1, out2, atol=max(atol, 1e-8), rtol=1e-5)
    assert torch.allclose(g1, g2, rtol=1e-4, atol=1e-5)


def test_scans_ragged_vs_oracle_and_reference(dev, oracle):
    g = load_golden("ragged_packed")
    pi = T(g["packed_info"], dev)
    gg = T(g["g"], dev)
    for kind, key in (("inclusive_sum", "x"), ("exclusive_sum", "x"), ("inclusive_prod", "xp"), ("exclusive_prod", "xp")):
        x = T(g[key], dev).requires_grad_(True)
        y = getattr(na, kind)(x, pi)
        (y * gg).sum().backward()
        assert_close(y, g[kind], atol=1e-5, rtol=1e-5, what=kind + " vs reference")
        assert_close(y, oracle.packed_scan(kind, g[key], g["packed_info"]), atol=1e-5, rtol=1e-5, what=kind + " vs oracle")
        assert_close(x.grad, g[kind + "_grad"], atol=2e-5, rtol=1e-4, what=kind + " grad")
    # docstring vectors scan.py:36-39 ...
    x = torch.arange(1.0, 10.0, device=dev)
    p3 = torch.tensor([[0, 2], [2, 3], [5, 4]], device=dev)
    assert na.inclusive_sum(x, p3).tolist() == [1, 3, 3, 7, 12, 6, 13, 21, 30]
    assert na.exclusive_sum(x, p3).tolist() == [0, 1, 0, 3, 7, 0, 6, 13, 21]
    assert na.inclusive_prod(x, p3).tolist() == [1, 2, 3, 12, 60, 6, 42, 336, 3024]
    assert na.exclusive_prod(x, p3).tolist() == [1, 1, 1, 3, 12, 1, 6, 42, 336]


def test_scans_many_tiles_long_and_empty_rays(dev, oracle):
    rng = np.random.default_rng(9)
    lens = rng.integers(0, 70, size=20000)
    lens[100] = 9000          # spans > 4 ownership tiles
    lens[101] = 0
    lens[5000:5200] = 0       # a run of empty rays
    lens[-3:] = 0             # trailing empty rays
    starts = np.cumsum(lens) - lens
    pi = np.stack([starts, lens], -1).astype(np.int64)
    n = int(lens.sum())
    x = (rng.random(n) * 0.1).astype(np.float32)
    for kind in ("inclusive_sum", "exclusive_sum"):
        y = getattr(na, kind)(T(x, dev), T(pi, dev))
        ref = np.zeros(n, np.float64)
        for s, l in pi:
            c = np.cumsum(x[s:s + l].astype(np.float64))
            ref[s:s + l] = c if kind == "inclusive_sum" else c - x[s:s + l]
        assert_close(y, ref.astype(np.float32), atol=1e-6, rtol=1e-5, what=kind)
    xp = (0.97 + 0.03 * rng.random(n)).astype(np.float32)
    y = na.exclusive_prod(T(xp, dev), T(pi, dev))
    assert_close(y, oracle.exclusive_prod(xp, pi), atol=1e-30, rtol=2e-5)


def test_scans_generic_chunks_and_normalize(dev, oracle):
    # overlapping / unordered / gapped chunks: only the reference-style per-ray kernel applies
    x = np.random.default_rng(1).random(100).astype(np.float32)
    pi = np.array([[50, 10], [0, 20], [15, 30], [90, 0], [99, 1]], np.int64)
    for kind in ("inclusive_sum", "exclusive_sum", "inclusive_prod", "exclusive_prod"):
        y = getattr(na, kind)(T(x, dev), T(pi, dev)).cpu().numpy()
        ref = oracle.packed_scan(kind, x, pi)
        for s, l in pi[[0, 4]]:  # chunks not overwritten by an overlapping one
            assert np.allclose(y[s:s + l], ref[s:s + l], rtol=1e-5)
    pi2 = np.array([[0, 33], [33, 0], [33, 67]], np.int64)
    for kind in ("inclusive_sum", "exclusive_sum"):
        y = getattr(na, kind)(T(x, dev), T(pi2, dev), normalize=True)
        assert_close(y, oracle.packed_scan(kind, x, pi2, normalize=True), atol=1e-6, rtol=1e-5)


def test_scans_unaligned_views_and_empty(dev, oracle):
    rng = np.random.default_rng(2)
    lens = rng.integers(0, 50, size=300)
    pi = np.stack([np.cumsum(lens) - lens, lens], -1).astype(np.int64)
    n = int(lens.sum())
    x = rng.random(n + 1).astype(np.float32)
    xt = T(x, dev)[1:]                     # 4-byte aligned only -> scalar path
    assert xt.data_ptr() % 16 != 0
    y = na.exclusive_sum(xt, T(pi, dev))
    assert_close(y, oracle.exclusive_sum(x[1:], pi), atol=1e-5, rtol=1e-5)
    e = na.inclusive_sum(torch.zeros(0, device=dev), torch.zeros((3, 2), dtype=torch.long, device=dev))
    assert e.shape == (0,)


# ----------------------------------------------------------------------------- volrend
def test_rendering_reference_vectors(dev):
    ri = torch.tensor([0, 2, 2, 2, 2], dtype=torch.int64, device=dev)
    alphas = torch.tensor([0.4, 0.3, 0.8, 0.8, 0.5], device=dev)
    # tests/test_rendering.py:8-34
    vis = na.render_visibility_from_alpha(alphas, ray_indices=ri, early_stop_eps=0.03, alpha_thre=0.0)
    assert vis.tolist() == [True, True, True, True, False] and vis.dtype == torch.bool
    vis = na.render_visibility_from_alpha(alphas, ray_indices=ri, early_stop_eps=0.05, alpha_thre=0.35)
    assert vis.tolist() == [True, False, True, True, False]
    # tests/test_rendering.py:38-57
    w, _ = na.render_weight_from_alpha(alphas, ray_indices=ri, n_rays=3)
    assert torch.allclose(w, torch.tensor([0.4, 0.3, 0.7 * 0.8, 0.14 * 0.8, 0.028 * 0.5], device=dev))
    # tests/test_rendering.py:61-83 density == alpha formulation
    torch.manual_seed(0)
    sig = torch.rand(5, device=dev); ts = torch.rand(5, device=dev); te = torch.rand(5, device=dev) + 1.0
    w1, _, _ = na.render_weight_from_density(ts, te, sig, ray_indices=ri, n_rays=3)
    w2, _ = na.render_weight_from_alpha(1.0 - torch.exp(-sig * (te - ts)), ray_indices=ri, n_rays=3)
    assert torch.allclose(w1, w2)
    # tests/test_rendering.py:87-106
    vals = torch.rand((5, 2), device=dev)
    acc = na.accumulate_along_rays(alphas, values=vals, ray_indices=ri, n_rays=3)
    assert acc.shape == (3, 2) and torch.allclose(acc[0], alphas[0] * vals[0]) and (acc[1] == 0).all()
    assert torch.allclose(acc[2], (alphas[1:, None] * vals[1:]).sum(0))
    # docstrings volrend.py:246-253, 403-409
    a7 = torch.tensor([0.4, 0.8, 0.1, 0.8, 0.1, 0.0, 0.9], device=dev)
    r7 = torch.tensor([0, 0, 0, 1, 1, 2, 2], device=dev)
    assert torch.allclose(na.render_transmittance_from_alpha(a7, ray_indices=r7), torch.tensor([1.0, 0.6, 0.12, 1.0, 0.2, 1.0, 1.0], device=dev))
    assert na.render_visibility_from_alpha(a7, ray_indices=r7, early_stop_eps=0.3, alpha_thre=0.2).tolist() == \
        [True, True, False, True, False, False, True]


def test_grads_reference_vectors(dev):
    # tests/test_rendering.py:110-193 through all six API spellings
    ri = torch.tensor([0, 2, 2, 2, 2], dtype=torch.int64, device=dev)
    pi = torch.tensor([[0, 1], [1, 0], [1, 4]], dtype=torch.long, device=dev)
    sig = torch.tensor([0.4, 0.8, 0.1, 0.8, 0.1], device=dev, requires_grad=True)
    ts = torch.rand_like(sig); te = ts + 1.0
    w_ref = torch.tensor([0.3297, 0.5507, 0.0428, 0.2239, 0.0174], device=dev)
    g_ref = torch.tensor([0.6703, 0.1653, 0.1653, 0.1653, 0.1653], device=dev)

    def check(weights):
        weights.sum().backward()
        g = sig.grad.clone(); sig.grad.zero_()
        assert torch.allclose(w_ref, weights, atol=1e-4) and torch.allclose(g_ref, g, atol=1e-4)

    for kw in (dict(ray_indices=ri, n_rays=3), dict(packed_info=pi, n_rays=3)):
        tr, _ = na.render_transmittance_from_density(ts, te, sig, **kw)
        check(tr * (1.0 - torch.exp(-sig * (te - ts))))
        check(na.render_weight_from_density(ts, te, sig, **kw)[0])
        check(na.render_weight_from_alpha(1.0 - torch.exp(-sig * (te - ts)), **kw)[0])


def test_volrend_ragged_vs_reference_and_oracle(dev, oracle):
    g = load_golden("ragged_packed")
    pi, ri = T(g["packed_info"], dev), T(g["ray_indices"], dev)
    n_rays = pi.shape[0]
    ts, te = T(g["ts"], dev), T(g["te"], dev)
    gw, gt, ga = T(g["gw"], dev), T(g["gt"], dev), T(g["ga"], dev)
    for kw in (dict(packed_info=pi), dict(ray_indices=ri, n_rays=n_rays)):
        sig = T(g["sig"], dev).requires_grad_(True)
        w, tr, al = na.render_weight_from_density(ts, te, sig, **kw)
        (w * gw + tr * gt + al * ga).sum().backward()
        assert_close(w, g["rwd_w"], what="w"); assert_close(tr, g["rwd_t"], what="T"); assert_close(al, g["rwd_a"], what="alpha")
        assert_close(sig.grad, g["rwd_gsig"], atol=2e-5, rtol=1e-4, what="grad sigma")
    ow, ot, oa = oracle.render_weight_from_density(g["ts"], g["te"], g["sig"], g["packed_info"])
    assert_close(w, ow); assert_close(tr, ot); assert_close(al, oa)
    # transmittance-only entry point, prefix_trans
    tr2, al2 = na.render_transmittance_from_density(ts, te, T(g["sig"], dev), packed_info=pi)
    assert_close(tr2, g["rwd_t"]); assert_close(al2, g["rwd_a"])
    wp, tp, _ = na.render_weight_from_density(ts, te, T(g["sig"], dev), packed_info=pi, prefix_trans=T(g["pref"], dev))
    assert_close(wp, g["rwd_pref_w"]); assert_close(tp, g["rwd_pref_t"])
    # alpha path + gradient
    al = T(g["alph"], dev).requires_grad_(True)
    wa, ta = na.render_weight_from_alpha(al, packed_info=pi)
    (wa * gw + ta * gt).sum().backward()
    assert_close(wa, g["rwa_w"]); assert_close(ta, g["rwa_t"])
    assert_close(al.grad, g["rwa_galpha"], atol=2e-5, rtol=1e-4, what="grad alpha")
    assert_close(na.render_transmittance_from_alpha(T(g["alph"], dev), packed_info=pi), g["rwa_t"])
    # gradients w.r.t. t_starts / t_ends (the reference composition is differentiable there too)
    ts_g, te_g = ts.clone().requires_grad_(True), te.clone().requires_grad_(True)
    sg = T(g["sig"], dev)
    w3, _, _ = na.render_weight_from_density(ts_g, te_g, sg, packed_info=pi)
    (w3 * gw).sum().backward()
    B_ = oracle.render_weight_from_density_backward(g["ts"], g["te"], g["sig"], g["packed_info"], g["gw"])
    dt = (g["te"] - g["ts"]).astype(np.float64)
    g_te = np.where(dt != 0, B_ / np.where(dt != 0, dt, 1) * g["sig"], 0.0)
    assert_close(te_g.grad, g_te.astype(np.float32), atol=5e-4, rtol=1e-3, what="grad t_ends")
    assert_close(ts_g.grad, -g_te.astype(np.float32), atol=5e-4, rtol=1e-3, what="grad t_starts")
    # visibility (ties within 1e-5 of a threshold are platform-dependent: exp ulps)
    eps_t, thre = float(g["eps_t"]), float(g["thre"])
    vd = na.render_visibility_from_density(ts, te, T(g["sig"], dev), packed_info=pi, early_stop_eps=eps_t, alpha_thre=thre)
    va = na.render_visibility_from_alpha(T(g["alph"], dev), ray_indices=ri, n_rays=n_rays, early_stop_eps=eps_t, alpha_thre=thre)
    assert ((vd.cpu().numpy() == g["vis_d"]) | g["guard_d"]).all() and ((va.cpu().numpy() == g["vis_a"]) | g["guard_a"]).all()
    # accumulate + its gradients vs torch index_add_
    wt = T(g["rwd_w"], dev).requires_grad_(True)
    rgb = T(g["rgb"], dev).requires_grad_(True)
    acc = na.accumulate_along_rays(wt, rgb, ri, n_rays)
    go = torch.rand_like(acc)
    (acc * go).sum().backward()
    assert_close(acc, g["acc_rgb"], atol=1e-5)
    wt2, rgb2 = wt.detach().clone().requires_grad_(True), rgb.detach().clone().requires_grad_(True)
    ref = torch.zeros_like(acc).index_add_(0, ri, wt2[:, None] * rgb2)
    (ref * go).sum().backward()
    assert_close(wt.grad, wt2.grad, atol=1e-6); assert_close(rgb.grad, rgb2.grad, atol=1e-6)
    assert_close(na.accumulate_along_rays(wt.detach(), None, ri, n_rays), g["acc_w"], atol=1e-5)
    v7 = torch.rand((wt.numel(), 7), device=dev)                # D > 4: several channel groups
    assert_close(na.accumulate_along_rays(wt.detach(), v7, ri, n_rays),
                 torch.zeros((n_rays, 7), device=dev).index_add_(0, ri, wt.detach()[:, None] * v7), atol=1e-5)
    # in-place version, sorted and unsorted indices
    out = torch.ones((n_rays, 3), device=dev)
    na.volrend.accumulate_along_rays_(wt.detach(), rgb.detach(), ri, out)
    assert_close(out, g["acc_rgb"] + 1.0, atol=1e-5)
    perm = torch.randperm(wt.numel(), device=dev)
    out2 = torch.zeros((n_rays, 3), device=dev)
    na.volrend.accumulate_along_rays_(wt.detach()[perm], rgb.detach()[perm], ri[perm], out2)
    assert_close(out2, g["acc_rgb"], atol=1e-5)
    assert_close(na.accumulate_along_rays(wt.detach()[perm], rgb.detach()[perm], ri[perm], n_rays), g["acc_rgb"], atol=1e-5)


def test_rendering_composite(dev):
    g = load_golden("ragged_packed")
    ri = T(g["ray_indices"], dev)
    n_rays = g["packed_info"].shape[0]
    ts, te = T(g["ts"], dev), T(g["te"], dev)
    rgb = T(g["rgb"], dev).requires_grad_(True)
    sig = T(g["sig"], dev).requires_grad_(True)
    colors, opac, depth, extras = na.rendering(ts, te, ri, n_rays, rgb_sigma_fn=lambda a, b, c: (rgb, sig),
                                               render_bkgd=T(g["bkgd"], dev))
    assert_close(colors, g["rend_colors"], atol=1e-5); assert_close(depth, g["rend_depths"], atol=1e-5, rtol=1e-4)
    assert_close(opac, g["acc_w"], atol=1e-5)
    assert set(extras) == {"weights", "alphas", "trans", "sigmas", "rgbs"}
    (colors.sum() + 0.3 * depth.sum() + opac.sum()).backward()
    # same thing through the reference's composition (torch autograd over our scans)
    rgb2, sig2 = rgb.detach().clone().requires_grad_(True), sig.detach().clone().requires_grad_(True)
    sdt = sig2 * (te - ts)
    tr = torch.exp(-na.exclusive_sum(sdt, T(g["packed_info"], dev)))
    w = tr * (1 - torch.exp(-sdt))
    z = lambda d: torch.zeros((n_rays, d), device=dev)
    c2 = z(3).index_add_(0, ri, w[:, None] * rgb2)
    o2 = z(1).index_add_(0, ri, w[:, None])
    d2 = z(1).index_add_(0, ri, w[:, None] * ((ts + te)[:, None] / 2.0)) / o2.clamp_min(torch.finfo(torch.float32).eps)
    c2 = c2 + T(g["bkgd"], dev) * (1.0 - o2)
    (c2.sum() + 0.3 * d2.sum() + o2.sum()).backward()
    assert_close(sig.grad, sig2.grad, atol=3e-5, rtol=1e-4); assert_close(rgb.grad, rgb2.grad, atol=1e-6)
    # empty input (volrend.py:91-93) and rgb_alpha_fn
    e = torch.zeros(0, device=dev)
    c, o, d, _ = na.rendering(e, e, torch.zeros(0, dtype=torch.long, device=dev), 4,
                              rgb_sigma_fn=lambda a, b, c: (None, None))
    assert c.shape == (4, 3) and (c == 0).all() and (o == 0).all()
    al = T(g["alph"], dev)
    c3, o3, _, ex = na.rendering(ts, te, ri, n_rays, rgb_alpha_fn=lambda a, b, c: (rgb.detach(), al))
    assert_close(o3, torch.zeros((n_rays, 1), device=dev).index_add_(0, ri, T(g["rwa_w"], dev)[:, None]), atol=1e-5)
    assert set(ex) == {"weights", "trans", "rgbs", "alphas"}


# ----------------------------------------------------------------------------- grid
def test_rendering_fused_pass_is_bit_identical_to_two_passes(dev, oracle):
    """The one-pass rendering (stage-A transmittance scan + stage-B accumulation scan, and the reverse pass
    that needs the ray id before its scan) against the two-pass composition, many tiles, ragged + empty rays,
    with gradients arriving at colours / opacity / depth AND at extras' weights / trans / alphas."""
    rng = np.random.default_rng(int(os.environ.get("NFA_FUZZ_SEED", "11")))
    cnts = rng.integers(0, 90, size=5000)
    cnts[rng.integers(0, 5000, size=40)] = 0
    cnts[17], cnts[4000] = 3001, 1500
    n_rays, n = cnts.size, int(cnts.sum())
    ri = torch.repeat_interleave(torch.arange(n_rays, device=dev), T(cnts, dev))
    ts_np = rng.uniform(0.0, 4.0, n).astype(np.float32)
    ts = T(ts_np, dev); te = T(ts_np + rng.uniform(1e-3, 0.05, n).astype(np.float32), dev)
    sig0, rgb0 = T(rng.uniform(0, 6, n).astype(np.float32), dev), T(rng.uniform(0, 1, (n, 3)).astype(np.float32), dev)
    gw = T(rng.normal(size=n).astype(np.float32), dev)
    res = []
    for fuse in (True, False):
        na.volrend.FUSE_RENDERING = fuse
        try:
            sig, rgb = sig0.clone().requires_grad_(True), rgb0.clone().requires_grad_(True)
            c, o, d, ex = na.rendering(ts, te, ri, n_rays, rgb_sigma_fn=lambda a, b, r: (rgb, sig),
                                       render_bkgd=torch.tensor([0.1, 0.5, 0.9], device=dev))
            loss = (c * c).sum() + 0.3 * d.sum() + (o * 0.7).sum() + (ex["weights"] * gw).sum() \
                + 0.2 * (ex["trans"] * gw.flip(0)).sum() + 0.1 * (ex["alphas"] ** 2).sum()
            loss.backward()
            res.append([c, o, d, ex["weights"], ex["trans"], ex["alphas"], sig.grad, rgb.grad])
        finally:
            na.volrend.FUSE_RENDERING = True
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # and against the oracle's composition (numpy fp32), forward
    w, tr, al = oracle.render_weight_from_density(ts_np, te.cpu().numpy(), sig0.cpu().numpy(),
                                                  packed_info=oracle.pack_info(ri.cpu().numpy(), n_rays))
    assert_close(res[0][3], w, atol=1e-5); assert_close(res[0][4], tr, atol=1e-5)
    acc = oracle.accumulate_along_rays(w, rgb0.cpu().numpy(), ri.cpu().numpy(), n_rays)
    opa = oracle.accumulate_along_rays(w, None, ri.cpu().numpy(), n_rays)
    assert_close(res[0][0], acc + np.array([0.1, 0.5, 0.9], np.float32) * (1 - opa), atol=2e-5)
    # only colours used (the common training loss): NULL gradients for everything else
    sig, rgb = sig0.clone().requires_grad_(True), rgb0.clone().requires_grad_(True)
    c, _, _, _ = na.rendering(ts, te, ri, n_rays, rgb_sigma_fn=lambda a, b, r: (rgb, sig))
    c.sum().backward()
    na.volrend.FUSE_RENDERING = False
    try:
        sig2, rgb2 = sig0.clone().requires_grad_(True), rgb0.clone().requires_grad_(True)
        c2, _, _, _ = na.rendering(ts, te, ri, n_rays, rgb_sigma_fn=lambda a, b, r: (rgb2, sig2))
        c2.sum().backward()
    finally:
        na.volrend.FUSE_RENDERING = True
    assert torch.equal(sig.grad, sig2.grad) and torch.equal(rgb.grad, rgb2.grad) and torch.equal(c, c2)


def test_batched_inputs_take_the_native_path(dev):
    """Batched (R, S) CUDA tensors run on the flat segmented engine with uniform segments (the reference
    composes torch.cumsum / cumprod, scan.py:42-44, volrend.py:203-206): same values as that composition
    (golden batched vectors come from the reference itself), forward and backward."""
    g = load_golden("batched_volrend")
    rng = np.random.default_rng(3)
    R, S = 1037, 48
    x = torch.from_numpy(rng.uniform(0.1, 1.0, (R, S)).astype(np.float32)).to(dev)
    assert_close(na.inclusive_sum(x), torch.cumsum(x.double(), -1).float(), atol=2e-5, rtol=1e-5)
    assert_close(na.exclusive_sum(x), (torch.cumsum(x.double(), -1) - x.double()).float(), atol=2e-5, rtol=1e-5)
    assert_close(na.inclusive_prod(x), torch.cumprod(x.double(), -1).float(), atol=1e-6, rtol=2e-5)
    assert_close(na.exclusive_prod(x[:, :12]), (torch.cumprod(x[:, :12].double(), -1) / x[:, :12].double()).float(),
                 atol=1e-6, rtol=2e-5)
    assert na.inclusive_sum(x[0]).shape == (S,) and na.exclusive_sum(x.view(17, 61, S)).shape == (17, 61, S)
    ts = torch.from_numpy(np.sort(rng.uniform(0, 4, (R, S + 1)).astype(np.float32), -1)).to(dev)
    t0, t1 = ts[:, :-1].contiguous(), ts[:, 1:].contiguous()
    sig = torch.from_numpy(rng.uniform(0, 8, (R, S)).astype(np.float32)).to(dev).requires_grad_(True)
    gw = torch.from_numpy(rng.normal(size=(R, S)).astype(np.float32)).to(dev)
    w, tr, al = na.render_weight_from_density(t0, t1, sig)
    ((w * gw).sum() + 0.5 * tr.sum()).backward()
    sig2 = sig.detach().clone().requires_grad_(True)
    sdt = sig2 * (t1 - t0)
    tr2 = torch.exp(-(torch.cumsum(sdt, -1) - sdt)); al2 = 1 - torch.exp(-sdt); w2 = tr2 * al2
    ((w2 * gw).sum() + 0.5 * tr2.sum()).backward()
    assert_close(w, w2, atol=1e-5); assert_close(tr, tr2, atol=1e-5); assert_close(al, al2, atol=1e-6)
    assert_close(sig.grad, sig2.grad, atol=3e-5, rtol=1e-4)
    a = torch.from_numpy(rng.uniform(0, 0.3, (R, S)).astype(np.float32)).to(dev).requires_grad_(True)
    w, tr = na.render_weight_from_alpha(a)
    (w * gw).sum().backward()
    a2 = a.detach().clone().requires_grad_(True)
    tr2 = torch.cumprod(torch.cat([torch.ones_like(a2[:, :1]), 1 - a2[:, :-1]], -1), -1)
    ((tr2 * a2) * gw).sum().backward()
    assert_close(w, tr2 * a2, atol=1e-5); assert_close(a.grad, a2.grad, atol=3e-5, rtol=1e-4)
    vis = na.render_visibility_from_density(t0, t1, sig.detach(), early_stop_eps=1e-2, alpha_thre=1e-3)
    ref = (tr.new_tensor(0) + torch.exp(-(torch.cumsum(sdt, -1) - sdt)) >= 1e-2) & (al2 >= 1e-3)
    assert vis.shape == (R, S) and (vis != ref).float().mean() < 1e-4
    # the reference's own batched vectors
    gts, gte = T(g["ts"], dev), T(g["te"], dev)
    w, tr, al = na.render_weight_from_density(gts, gte, T(g["sig"], dev))
    assert_close(w, g["w"], atol=1e-5); assert_close(tr, g["trans"], atol=1e-5); assert_close(al, g["alphas"], atol=1e-6)


def test_ray_aabb_intersect(dev, oracle):
    g = load_golden("ray_aabb")
    tm, tM, hit = na.ray_aabb_intersect(T(g["rays_o"], dev), T(g["rays_d"], dev), T(g["aabbs"], dev))
    assert (hit.cpu().numpy() == g["hits"]).all()
    assert np.allclose(tm.cpu().numpy(), g["t_mins"]) and np.allclose(tM.cpu().numpy(), g["t_maxs"])
    o = oracle.ray_aabb_intersect(g["rays_o"], g["rays_d"], g["aabbs"], 0.3, 1.1, -1.0)
    h = na.ray_aabb_intersect(T(g["rays_o"], dev), T(g["rays_d"], dev), T(g["aabbs"], dev), 0.3, 1.1, -1.0)
    for a, b in zip(h, o):
        assert (a.cpu().numpy() == b).all()  # bit-exact vs the oracle
    # mid-points of hits are inside the boxes (tests/test_grid.py:29-35)
    tmid = torch.clamp((tm + tM) / 2, min=0.0)
    pts = tmid[:, :, None] * T(g["rays_d"], dev)[:, None, :] + T(g["rays_o"], dev)[:, None, :]
    ab = T(g["aabbs"], dev)
    inside = ((pts >= ab[None, :, :3]) & (pts <= ab[None, :, 3:])).all(-1)
    assert (inside == hit).all()


def test_ray_events_equal_intersection_plus_stable_sort(dev, oracle):
    """nfa_ray_events (intersection + sort of the 2 G distances in registers) against the reference's composition
    (grid.py:156-162) on the native intersection, with a STABLE torch.sort: same values, same indices -- ties included
    (the +inf pairs of missed boxes) --, same hits; rays from inside, from outside, axis-parallel, missing every box."""
    from nerfacc_amd.grid import ray_events
    rng = np.random.default_rng(11)
    for G in (1, 2, 3, 4, 8):
        n = 5000
        o = rng.standard_normal((n, 3)).astype(np.float32) * 1.5
        d = rng.standard_normal((n, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        o[:500] *= 0.1                                   # inside the innermost box
        d[500:600, 1] = 0.0; d[600:650, :2] = 0.0; d[600:650, 2] = 1.0   # axis-parallel (1 / d = inf)
        o[650:700] = 50.0                                # far away, most boxes missed
        base = np.array([-1, -1, -1, 1, 1, 1], dtype=np.float32)
        aabbs = np.stack([base * 2.0 ** k for k in range(G)]).astype(np.float32)
        ro, rd, ab = T(o, dev), T(d, dev), T(aabbs, dev)
        ts, ti, hits = ray_events(ro, rd, ab)
        t_mins, t_maxs, h_ref = na.ray_aabb_intersect(ro, rd, ab)
        cat = torch.cat([t_mins, t_maxs], -1)
        ts_ref, ti_ref = torch.sort(cat, dim=-1, stable=True)
        assert ts.dtype == torch.float32 and ti.dtype == torch.int64 and hits.dtype == torch.bool
        assert ts.shape == (n, 2 * G) and ti.shape == (n, 2 * G) and hits.shape == (n, G)
        assert torch.equal(hits, h_ref), G
        assert torch.equal(ts.view(torch.int32), ts_ref.view(torch.int32)), G      # bit for bit (inf included)
        assert torch.equal(ti, ti_ref), G
        om, oM, oh = oracle.ray_aabb_intersect(o, d, aabbs, -np.inf, np.inf, np.inf)
        ocat = np.concatenate([om, oM], -1)
        oi = np.argsort(ocat, axis=-1, kind="stable")
        assert (ti.cpu().numpy() == oi).all() and (ts.cpu().numpy() == np.take_along_axis(ocat, oi, -1)).all(), G
    # more boxes than the native pass takes: the composition
    G = 9
    aabbs = np.stack([base * 1.5 ** k for k in range(G)]).astype(np.float32)
    ts, ti, hits = ray_events(ro, rd, T(aabbs, dev))
    assert ts.shape == (n, 2 * G) and (ts[:, 1:] >= ts[:, :-1]).all()
    # an empty batch
    ts, ti, hits = ray_events(ro[:0], rd[:0], ab)
    assert ts.shape == (0, 2 * ab.shape[0]) and ti.shape == ts.shape and hits.shape == (0, ab.shape[0])


def _cmp_traversal(res, ref):
    iv, sm, term = res
    riv, rsm, rterm = ref
    assert (iv.packed_info.cpu().numpy() == riv["packed_info"]).all(), "interval packed_info"
    assert (sm.packed_info.cpu().numpy() == rsm["packed_info"]).all(), "sample packed_info"
    assert (iv.vals.cpu().numpy() == riv["vals"]).all(), "interval vals (bit-exact)"
    assert (iv.is_left.cpu().numpy() == riv["is_left"]).all() and (iv.is_right.cpu().numpy() == riv["is_right"]).all()
    assert (iv.ray_indices.cpu().numpy() == riv["ray_indices"]).all()
    assert (sm.vals.cpu().numpy() == rsm["vals"]).all() and (sm.ray_indices.cpu().numpy() == rsm["ray_indices"]).all()
    assert (sm.is_valid.cpu().numpy() == rsm["is_valid"]).all()
    assert (term.cpu().numpy() == rterm).all(), "terminate planes"


def test_traverse_grids_multilevel_bit_exact(dev, oracle):
    g = load_golden("traversal")
    binaries = np.unpackbits(g["a_binaries"]).astype(bool).reshape(4, 32, 32, 32)
    o, d, ab = g["a_rays_o"], g["a_rays_d"], g["a_aabbs"]
    res = na.traverse_grids(T(o, dev), T(d, dev), T(binaries, dev), T(ab, dev))
    ref = oracle.traverse_grids(o, d, binaries, ab)
    _cmp_traversal(res, ref)
    assert (res[0].packed_info.cpu().numpy() == g["a64_iv_packed"]).all()  # committed golden counts
    assert (res[1].packed_info.cpu().numpy() == g["a64_sm_packed"]).all()
    # the reference's property test (tests/test_grid.py:57-68) on the first 8 rays
    iv, sm, _ = res
    ts, te = iv.vals[iv.is_left], iv.vals[iv.is_right]
    ri = sm.ray_indices
    sel8 = ri < 8
    pos = T(o, dev)[ri] + T(d, dev)[ri] * (ts + te)[:, None] / 2.0
    occ, selector = na.grid._query(pos[sel8], T(binaries, dev), T(np.array([-1, -1, -1, 1, 1, 1], np.float32), dev))
    assert selector.all() and occ.float().mean() > 0.9999


@pytest.mark.parametrize("tag", ["cfg1", "cfg1b", "cone", "percell"])
def test_traverse_grids_seeded_cases_bit_exact(dev, oracle, tag):
    g = load_golden("traversal")
    o, d, b, ab, nearp, step, cone = seeded_case(g[f"{tag}_params"])
    res = na.traverse_grids(T(o, dev), T(d, dev), T(b, dev), T(ab, dev), near_planes=T(nearp, dev), step_size=step,
                            cone_angle=cone)
    ref = oracle.traverse_grids(o, d, b, ab, near_planes=nearp, step_size=step, cone_angle=cone)
    _cmp_traversal(res, ref)
    assert res[1].vals.numel() == int(g[f"{tag}_M"]) and res[0].vals.numel() == int(g[f"{tag}_E"])
    # explicit intersections (the non-fused kernel) give the same thing for one grid
    if b.shape[0] == 1:
        tm, tM, hits = na.ray_aabb_intersect(T(o, dev), T(d, dev), T(ab, dev))
        t_sorted, t_idx = torch.sort(torch.cat([tm, tM], -1), -1)
        res2 = na.traverse_grids(T(o, dev), T(d, dev), T(b, dev), T(ab, dev), near_planes=T(nearp, dev),
                                 step_size=step, cone_angle=cone, t_sorted=t_sorted, t_indices=t_idx, hits=hits)
        _cmp_traversal(res2, ref)


def test_traverse_near_far_planes(dev, oracle):
    # tests/test_grid.py:135-159
    ro = torch.tensor([[-1.0, 0.0, 0.0]], device=dev)
    rd = torch.tensor([[1.0, 0.01, 0.01]], device=dev); rd = rd / rd.norm(dim=-1, keepdim=True)
    binaries = torch.ones((1, 1, 1, 1), dtype=torch.bool, device=dev)
    aabbs = torch.tensor([[0.0, 0.0, 0.0, 1.0, 1.0, 1.0]], device=dev)
    near, far, step = torch.tensor([1.2], device=dev), torch.tensor([1.5], device=dev), 0.05
    iv, sm, _ = na.traverse_grids(ro, rd, binaries, aabbs, step_size=step, near_planes=near, far_planes=far)
    assert iv.vals.numel() > 0
    assert (iv.vals >= (near - step / 2)).all() and (iv.vals <= (far + step / 2)).all()
    g = load_golden("traversal")
    assert (iv.vals.cpu().numpy() == g["c_vals"]).all() and (iv.is_left.cpu().numpy() == g["c_left"]).all()


def test_traverse_test_mode_over_allocate(dev, oracle):
    # tests/test_grid.py:72-131 plus bit-exactness of every chunk against the oracle
    g = load_golden("traversal")
    binaries = np.unpackbits(g["a_binaries"]).astype(bool).reshape(4, 32, 32, 32)
    o, d, ab = g["a_rays_o"], g["a_rays_d"], g["a_aabbs"]
    to, td, tb, tab = T(o, dev), T(d, dev), T(binaries, dev), T(ab, dev)
    n = o.shape[0]
    one = na.traverse_grids(to, td, tb, tab)
    tp = mask = None
    otp = omask = None
    total = torch.zeros(n, dtype=torch.long, device=dev)
    for it in range(8):
        iv, sm, tp = na.traverse_grids(to, td, tb, tab, near_planes=tp, traverse_steps_limit=4000, over_allocate=True,
                                       rays_mask=mask)
        oiv, osm, otp = oracle.traverse_grids(o, d, binaries, ab, near_planes=otp, traverse_steps_limit=4000,
                                              over_allocate=True, rays_mask=omask)
        _cmp_traversal((iv, sm, tp), (oiv, osm, otp))
        assert iv.vals.numel() == (int(mask.sum()) if mask is not None else n) * 8000
        mask = sm.packed_info[:, 1] == 4000
        omask = osm["packed_info"][:, 1] == 4000
        total += sm.packed_info[:, 1]
        assert (sm.ray_indices[sm.is_valid].cpu().numpy() == osm["ray_indices"][osm["is_valid"]]).all()
        if not mask.any():
            break
    assert (~mask).all()
    assert ((total - one[1].packed_info[:, 1]).abs() <= 2).all()  # see oracle/gen_golden.py (resume re-seeds the DDA)


def test_sampling_bit_exact_and_properties(dev, oracle):
    # tests/test_grid.py:163-203 recipe, compared with the oracle's restatement of occ_grid.py:85-221
    rng = np.random.default_rng(42)
    n_rays, levels, res, step = 256, 4, 32, 0.01
    o = (rng.random((n_rays, 3)) * 2 - 1).astype(np.float32)
    d = rng.random((n_rays, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    b = rng.random((levels, res, res, res)) > 0.5
    t_min = rng.random(n_rays).astype(np.float32); t_max = (t_min + rng.random(n_rays)).astype(np.float32)
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=levels).to(dev)
    est.binaries = T(b, dev)
    ri, ts, te = est.sampling(T(o, dev), T(d, dev), near_plane=0.15, far_plane=0.85, t_min=T(t_min, dev),
                              t_max=T(t_max, dev), render_step_size=step)
    assert (ts >= (T(t_min, dev)[ri] - step / 2)).all() and (te <= (T(t_max, dev)[ri] + step / 2)).all()
    ab = est.aabbs.cpu().numpy()
    ori, ots, ote = oracle.occgrid_sampling(o, d, b, ab, near_plane=0.15, far_plane=0.85, t_min=t_min, t_max=t_max,
                                            render_step_size=step)
    assert (ri.cpu().numpy() == ori).all() and (ts.cpu().numpy() == ots).all() and (te.cpu().numpy() == ote).all()

    # with a density callback: visibility + compaction.  sigma is an exactly representable function
    # of the ray index, so only exp() ulps can differ; samples within the guard band may flip.
    def sigma_np(ts_, te_, ri_):
        return (1.0 + (ri_ % 7)).astype(np.float32) * 8.0

    est.occs.fill_(0.5)
    ri2, ts2, te2 = est.sampling(T(o, dev), T(d, dev), sigma_fn=lambda a, b_, c: (1.0 + (c % 7)).float() * 8.0,
                                 render_step_size=step, early_stop_eps=1e-2, alpha_thre=0.1)
    (ori2, ots2, ote2), (fri, fts, fte, fpi) = oracle.occgrid_sampling(
        o, d, b, ab, sigma_fn=sigma_np, render_step_size=step, early_stop_eps=1e-2, alpha_thre=0.1, occs_mean=0.5,
        return_all=True)
    tr, al = oracle.render_transmittance_from_density(fts, fte, sigma_np(fts, fte, fri), fpi)
    thre = min(0.1, 0.5)
    guard = (np.abs(tr - 1e-2) < 1e-6) | (np.abs(al - thre) < 1e-6)
    if not guard.any():
        assert (ri2.cpu().numpy() == ori2).all() and (ts2.cpu().numpy() == ots2).all() and (te2.cpu().numpy() == ote2).all()
    else:  # every non-guarded sample must agree
        keep = ~guard
        vis = oracle.render_visibility_from_density(fts, fte, sigma_np(fts, fte, fri), fpi, 1e-2, thre)
        exp = set(zip(fri[vis & keep].tolist(), fts[vis & keep].tolist()))
        got = set(zip(ri2.cpu().numpy().tolist(), ts2.cpu().numpy().tolist()))
        assert exp <= got and len(got) - len(exp) <= int(guard.sum())
    assert (ri2[1:] >= ri2[:-1]).all()
    # alpha_fn spelling
    ri3, ts3, te3 = est.sampling(T(o, dev), T(d, dev), alpha_fn=lambda a, b_, c: torch.full_like(a, 0.05),
                                 render_step_size=step, early_stop_eps=0.5, alpha_thre=0.0)
    cnt = torch.bincount(ri3, minlength=n_rays)
    assert cnt.max() <= 14  # 0.95^k >= 0.5  =>  k <= 13 samples kept per ray (+1)
    # empty result
    est.binaries = torch.zeros_like(est.binaries)
    r0, t0, t1 = est.sampling(T(o, dev), T(d, dev), sigma_fn=lambda a, b_, c: a, render_step_size=step)
    assert r0.numel() == 0 and t0.numel() == 0 and r0.dtype == torch.int64


@pytest.mark.parametrize("tag", ["cfg1", "cfg1b"])
def test_sampler_run_length_path_bit_exact(dev, oracle, tag):
    """OccGridEstimator.sampling's traversal (brick grid + runs + parallel expansion, with the
    serial fill for rays that have more than 32 runs) == oracle's two-pass traversal, bit for bit."""
    g = load_golden("traversal")
    o, d, b, ab, nearp, step, cone = seeded_case(g[f"{tag}_params"])
    ri, ts, te, pi = na.grid._traverse_samples(T(o, dev), T(d, dev), T(b, dev), T(ab, dev), T(nearp, dev),
                                               torch.full((o.shape[0],), 1e10, device=dev), step, cone)
    iv, sm, _ = oracle.traverse_grids(o, d, b, ab, near_planes=nearp, far_planes=np.full(o.shape[0], 1e10, np.float32),
                                      step_size=step, cone_angle=cone)
    assert (pi.cpu().numpy() == sm["packed_info"]).all()
    assert (ri.cpu().numpy() == sm["ray_indices"]).all()
    assert (ts.cpu().numpy() == iv["vals"][iv["is_left"]]).all() and (te.cpu().numpy() == iv["vals"][iv["is_right"]]).all()
    assert ri.numel() == int(g[f"{tag}_M"])


def test_sampler_odd_resolution_global_brick_mask(dev, oracle):
    """Resolution not a multiple of 4 and a brick mask too large for LDS (65^3 bricks)."""
    rng = np.random.default_rng(5)
    R, res = 2048, 258
    o = rng.standard_normal((R, 3)).astype(np.float32) * 0.7
    d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    b = rng.random((1, res, res - 3, res + 1)) < 0.03
    ab = np.array([[-1, -1.5, -1, 1, 1, 1.25]], np.float32)
    near = np.zeros(R, np.float32); far = np.full(R, 1e10, np.float32)
    ri, ts, te, pi = na.grid._traverse_samples(T(o, dev), T(d, dev), T(b, dev), T(ab, dev), T(near, dev), T(far, dev),
                                               3e-3, 0.0)
    iv, sm, _ = oracle.traverse_grids(o, d, b, ab, near_planes=near, far_planes=far, step_size=3e-3)
    assert (pi.cpu().numpy() == sm["packed_info"]).all() and (ri.cpu().numpy() == sm["ray_indices"]).all()
    assert (ts.cpu().numpy() == iv["vals"][iv["is_left"]]).all() and (te.cpu().numpy() == iv["vals"][iv["is_right"]]).all()


def test_sampler_cone_angle_bit_exact(dev, oracle):
    """Distance-dependent steps (cone_angle > 0) in the sampler == the oracle's serial two-pass traversal, bit for bit:
    single grid with in-kernel intersection, nested levels with unrelated rays, resolutions that are not multiples of
    4, cells much larger than the step (closed-form skips), zero direction components, per-ray near / far planes.
    The sampler's path is run records from the count pass + the recurrence expansion (nfa_traverse_cone_runs,
    nfa_expand_cone_runs), with the serial fill for rays of more than 32 records; it is also compared with the serial
    count + fill passes (ray_indices from nfa_fill_ray_indices, also for a ray of millions of samples)."""
    g = load_golden("traversal")
    o, d, b, ab, nearp, step, cone = seeded_case(g["cone_params"])
    assert cone > 0
    far = np.full(o.shape[0], 1e10, np.float32)

    def check(o, d, b, ab, near, far, step, cone):
        ri, ts, te, pi = na.grid._traverse_samples(T(o, dev), T(d, dev), T(b, dev), T(ab, dev), T(near, dev), T(far, dev),
                                                   step, cone)
        iv, sm, _ = oracle.traverse_grids(o, d, b, ab, near_planes=near, far_planes=far, step_size=step, cone_angle=cone)
        assert (pi.cpu().numpy() == sm["packed_info"]).all() and (ri.cpu().numpy() == sm["ray_indices"]).all()
        assert (ts.cpu().numpy() == iv["vals"][iv["is_left"]]).all() and (te.cpu().numpy() == iv["vals"][iv["is_right"]]).all()
        return ri.numel()

    assert check(o, d, b, ab, nearp, far, step, cone) == int(g["cone_M"])
    rng = np.random.default_rng(int(os.environ.get("NFA_FUZZ_SEED", "77")))            # soak runs: other seeds, more cases
    for case in range(int(os.environ.get("NFA_FUZZ_CASES", "8"))):
        levels = int(rng.integers(1, 5))
        res = [int(rng.choice([8, 16, 30, 50]))] * 3 if case % 2 else [int(rng.choice([12, 32, 66])), 32, int(rng.choice([20, 48]))]
        if case == 5:
            res = [258, 255, 259]; levels = 1
        if case == 3:
            res = [8, 8, 8]; levels = 2                           # cells of 250+ steps: closed-form skips
        n_rays = int(rng.integers(300, 1500))
        b = rng.random((levels, *res)) < (0.03 if case == 3 else float(rng.choice([0.03, 0.2, 0.6])))
        o = (rng.random((n_rays, 3)) * 3 - 1.5).astype(np.float32) * (0.3 if case % 2 else 1.0)
        d = rng.standard_normal((n_rays, 3)).astype(np.float32)
        d[rng.random(n_rays) < 0.1, int(rng.integers(0, 3))] = 0.0
        d /= np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-6)
        est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=levels).to(dev)
        ab = est.aabbs.cpu().numpy()
        near = (rng.random(n_rays) * 0.5).astype(np.float32) if case % 3 == 1 else np.full(n_rays, 0.05, np.float32)
        far = (near + 0.5 + rng.random(n_rays) * 6).astype(np.float32) if case % 3 == 1 else np.full(n_rays, 1e10, np.float32)
        step = 1e-3 if case == 3 else float(rng.choice([2e-3, 5e-3, 0.02]))
        cone = float(rng.choice([0.004, 0.02, 1e-3]))
        assert check(o, d, b, ab, near, far, step, cone) > 0, case
    # rays with more than 32 records (checkerboard: a chain per cell) take the serial fill; and a training-sized batch
    # of unrelated rays through 4 levels, run records + expansion against the serial count + fill passes
    ii = np.indices((48, 48, 48)).sum(0)
    b = np.broadcast_to(ii % 2 == 0, (2, 48, 48, 48)).copy()
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=48, levels=2).to(dev)
    o = (rng.random((700, 3)) * 2 - 1).astype(np.float32)
    d = rng.standard_normal((700, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    assert check(o, d, b, est.aabbs.cpu().numpy(), np.full(700, 0.05, np.float32), np.full(700, 1e10, np.float32), 4e-3, 0.004) > 0
    n_rays, levels, res = 40000, 4, 64
    b = T(rng.random((levels, res, res, res)) < 0.1, dev)
    o = T((rng.random((n_rays, 3)) - 0.5).astype(np.float32), dev)
    d = rng.standard_normal((n_rays, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=levels).to(dev)
    args = (o, T(d, dev), b, est.aabbs, torch.full((n_rays,), 0.2, device=dev), torch.full((n_rays,), 1e10, device=dev),
            2e-3, 0.004)
    assert na.grid.CONE_RUNS
    new = na.grid._traverse_samples(*args, return_terminate=True)
    na.grid.CONE_RUNS = False
    try:
        old = na.grid._traverse_samples(*args, return_terminate=True)
    finally:
        na.grid.CONE_RUNS = True
    assert new[0].numel() > 1_000_000
    for x, y in zip(new, old):
        assert torch.equal(x, y)
    # nfa_fill_ray_indices alone: ragged counts with empty rays, and a window of more than 2^27 samples
    from nerfacc_amd import _backend as B
    for counts in (rng.integers(0, 700, 5000) * (rng.random(5000) < 0.7), np.array([3, 0, (1 << 27) + 5, 7, 0, 2])):
        counts = counts.astype(np.int64)
        starts = np.cumsum(counts) - counts
        pi = T(np.stack([starts, counts], -1), dev)
        out = torch.empty(int(counts.sum()), dtype=torch.int64, device=dev)
        B.call("nfa_fill_ray_indices", len(counts), B.ptr(pi), B.ptr(out), B.stream())
        ref = torch.repeat_interleave(torch.arange(len(counts), device=dev), T(counts, dev))
        assert torch.equal(out, ref)
        del out, ref


# ----------------------------------------------------------------------------- pdf
def test_importance_sampling_and_searchsorted(dev, oracle):
    g = load_golden("pdf")
    for tag in "abc":
        iv = na.RayIntervals(vals=T(g[f"{tag}_vals"], dev))
        out_iv, out_sm = na.importance_sampling(iv, T(g[f"{tag}_cdfs"], dev), int(g[f"{tag}_S"]), False)
        assert_close(out_iv.vals, g[f"{tag}_twin_edges"], atol=1e-4, rtol=0)       # tests/test_pdf.py:93-94
        assert_close(out_sm.vals, g[f"{tag}_twin_centres"], atol=1e-4, rtol=0)
        assert_close(out_iv.vals, g[f"{tag}_oracle_edges"], atol=1e-6, rtol=1e-6)
        assert_close(out_sm.vals, g[f"{tag}_oracle_centres"], atol=1e-6, rtol=1e-6)
    # docstring examples pdf.py:108-120 (packed input), :40-56
    iv = na.RayIntervals(vals=torch.tensor([0.0, 1.0, 0.0, 1.0, 2.0], device=dev),
                         packed_info=torch.tensor([[0, 2], [2, 3]], device=dev))
    o_iv, o_sm = na.importance_sampling(iv, torch.tensor([0.0, 0.5, 0.0, 0.5, 1.0], device=dev), 2)
    assert torch.allclose(o_iv.vals, torch.tensor([[0, 0.5, 1.0], [0, 1.0, 2.0]], device=dev))
    assert torch.allclose(o_sm.vals, torch.tensor([[0.25, 0.75], [0.5, 1.5]], device=dev))
    vals = na.RayIntervals(vals=torch.tensor([0.5, 1.5, 2.5], device=dev), packed_info=torch.tensor([[0, 1], [1, 2]], device=dev))
    l, r = na.searchsorted(iv, vals)
    assert l.tolist() == [0, 3, 3] and r.tolist() == [1, 4, 4]
    # batched searchsorted vs torch (tests/test_pdf.py:46-62) and vs the oracle
    key, query = T(g["loss_k_vals"], dev), T(g["loss_q_vals"], dev)
    il, ir = na.searchsorted(na.RayIntervals(vals=key), na.RayIntervals(vals=query))
    ref = torch.clamp(torch.searchsorted(key, query, right=True), 0, key.shape[-1] - 1)
    assert (ir == ref).all() and (il.cpu().numpy() == g["loss_ids_left"]).all()
    # RaySamples as query (broken in the reference, data_specs.py:57)
    il2, _ = na.searchsorted(na.RayIntervals(vals=key), na.RaySamples(vals=query))
    assert (il2 == il).all()
    # _pdf_loss == reference's _lossfun_outer where both are defined (tests/test_pdf.py:98-127)
    from nerfacc_amd.estimators.prop_net import _pdf_loss
    loss = _pdf_loss(na.RayIntervals(vals=query), T(g["loss_q_cdfs"], dev), na.RayIntervals(vals=key), T(g["loss_k_cdfs"], dev))
    inside = g["loss_inside"]
    assert np.allclose(loss.cpu().numpy()[inside], g["loss_ref"][inside], atol=1e-4)
    # long rows (S > 64) and stratified determinism / statistics
    rng = np.random.default_rng(8)
    v = np.sort(rng.random((33, 300)), -1).astype(np.float32); c = np.sort(rng.random((33, 300)), -1).astype(np.float32)
    o_iv, o_sm = na.importance_sampling(na.RayIntervals(vals=T(v, dev)), T(c, dev), 150)
    e_iv, e_sm = oracle.importance_sampling(v, c, 150)
    assert_close(o_iv.vals, e_iv, atol=1e-6, rtol=1e-6); assert_close(o_sm.vals, e_sm, atol=1e-6, rtol=1e-6)
    torch.manual_seed(123)
    gen = torch.cuda.default_generators[0]
    seed, off = gen.initial_seed(), gen.get_offset()
    s_iv, s_sm = na.importance_sampling(na.RayIntervals(vals=T(v, dev)), T(c, dev), 16, stratified=True)
    assert gen.get_offset() == off + 4                                  # pdf.cu:376-383
    e_iv, e_sm = oracle.importance_sampling(v, c, 16, True, seed=seed, offset=off)
    assert_close(s_sm.vals, e_sm, atol=1e-6, rtol=1e-6)                # same Philox stream as the oracle
    s2_iv, _ = na.importance_sampling(na.RayIntervals(vals=T(v, dev)), T(c, dev), 16, stratified=True)
    assert not torch.equal(s_iv.vals, s2_iv.vals)                       # generator advanced
    # the short-row kernel (S <= 64, CDF rows staged): enough rays that a wave's block of rays shares out its Philox draws
    # over the lanes (R >= 65536 for 8 rays per group), a ragged last block, S == 1, S == lanes
    R2 = 70001
    v2 = np.sort(rng.random((R2, 9)), -1).astype(np.float32); c2 = np.sort(rng.random((R2, 9)), -1).astype(np.float32)
    for S2, strat in ((8, True), (5, False), (64, True), (24, True), (33, False)):
        seed, off = gen.initial_seed(), gen.get_offset()
        r_iv, r_sm = na.importance_sampling(na.RayIntervals(vals=T(v2, dev)), T(c2, dev), S2, stratified=strat)
        e_iv, e_sm = oracle.importance_sampling(v2, c2, S2, strat, seed=seed, offset=off)
        assert_close(r_iv.vals, e_iv, atol=1e-6, rtol=1e-6); assert_close(r_sm.vals, e_sm, atol=1e-6, rtol=1e-6)
    r_iv, _ = na.importance_sampling(na.RayIntervals(vals=T(v2, dev)), T(c2, dev), 1)     # S == 1: the ray's whole range
    assert np.array_equal(r_iv.vals.cpu().numpy(), v2[:, [0, -1]])
    # s -> t mapping fused into the resampling: same intervals, and t rows bit-equal to the reference's tensor
    # expression (estimators/prop_net.py:215-229) on them, for both mappings, short and long rows, packed input too
    from nerfacc_amd.estimators.prop_net import _transform_stot
    for kind, lo, hi in (("uniform", 2.0, 6.0), ("lindisp", 0.05, 1e3), ("lindisp", 0.2, 7.3)):
        for S in (2, 16, 150):
            p_iv, p_sm = na.importance_sampling(na.RayIntervals(vals=T(v, dev)), T(c, dev), S)
            f_iv, f_sm, f_ts, f_te = na.importance_sampling(na.RayIntervals(vals=T(v, dev)), T(c, dev), S,
                                                            transform=(kind, lo, hi))
            assert torch.equal(p_iv.vals, f_iv.vals) and torch.equal(p_sm.vals, f_sm.vals)
            t_ref = _transform_stot(kind, p_iv.vals, lo, hi)
            assert f_ts.is_contiguous() and f_te.is_contiguous() and f_ts.shape == (33, S)
            assert torch.equal(f_ts, t_ref[:, :-1]) and torch.equal(f_te, t_ref[:, 1:])
    n_iv, n_sm = na.importance_sampling(na.RayIntervals(vals=T(v, dev)), T(c, dev), 16, need_samples=False)
    assert n_sm is None and torch.equal(n_iv.vals, na.importance_sampling(na.RayIntervals(vals=T(v, dev)), T(c, dev), 16)[0].vals)
    _, _, pk_ts, pk_te = na.importance_sampling(iv, torch.tensor([0.0, 0.5, 0.0, 0.5, 1.0], device=dev), 2,
                                                transform=("uniform", 1.0, 3.0))
    assert pk_ts.tolist() == [[1.0, 2.0], [1.0, 3.0]] and pk_te.tolist() == [[2.0, 3.0], [3.0, 5.0]]
    with pytest.raises(ValueError):
        na.importance_sampling(na.RayIntervals(vals=T(v, dev)), T(c, dev), 4, transform=("log", 1.0, 2.0))
    with pytest.raises(ValueError):
        na.importance_sampling(na.RayIntervals(vals=T(v, dev)), T(c, dev), 0)


def test_propnet_sampling_and_loss(dev):
    est = na.PropNetEstimator().to(dev)
    n_rays = 257
    fn = lambda ts, te: torch.exp(-((ts + te) * 0.5 - 4.0) ** 2) * 3.0
    ts, te = est.sampling([fn, fn], [64, 32], 16, n_rays, 2.0, 6.0, sampling_type="uniform")
    assert ts.shape == (n_rays, 16) and (te > ts).all() and (ts >= 2.0 - 1e-5).all() and (te <= 6.0 + 1e-5).all()
    assert (ts[:, 1:] == te[:, :-1]).all()
    # samples concentrate where the proposal density is high
    assert ((ts + te) * 0.5 - 4.0).abs().median() < 0.8
    ts2, te2 = est.sampling([fn], [48], 24, n_rays, 2.0, 6.0, sampling_type="lindisp", stratified=True)
    assert ts2.shape == (n_rays, 24) and (te2 > ts2).all()
    # training path: cached proposals + loss gradient reaches the proposal parameters
    p = torch.nn.Parameter(torch.tensor(3.0, device=dev))
    est2 = na.PropNetEstimator(optimizer=torch.optim.SGD([p], lr=1e-2)).to(dev)
    pfn = lambda ts, te: torch.exp(-((ts + te) * 0.5 - 4.0) ** 2) * p
    ts, te = est2.sampling([pfn], [64], 16, n_rays, 2.0, 6.0, sampling_type="uniform", requires_grad=True)
    sig = fn(ts, te)
    trans, _ = na.render_transmittance_from_density(ts, te, sig)
    loss = est2.update_every_n_steps(trans, requires_grad=True)
    assert np.isfinite(loss) and len(est2.prop_cache) == 0 and float(p) != 3.0


def test_propnet_sampling_vs_oracle(dev, oracle):
    """PropNetEstimator.sampling (ref estimators/prop_net.py:38-129) against the oracle's restatement of the level loop at
    the shape of BASELINE cfg 3 (2 -> 64 -> 16 edges per ray, and a two-proposal 64 -> 64 -> 16 variant): final
    (t_starts, t_ends) and every cached (intervals, cdfs) level within 1e-5 of the distance range."""
    n_rays = 4099
    off = torch.linspace(-0.6, 0.6, n_rays, device=dev)[:, None]
    fn_t = lambda ts, te: torch.exp(-((ts + te) * 0.5 - 4.0 - off) ** 2 * 2.0) * 3.0 + 0.05
    off_np = off.cpu().numpy()
    fn_np = lambda ts, te: (np.exp(-((ts + te) * np.float32(0.5) - np.float32(4.0) - off_np) ** 2 * np.float32(2.0)) * np.float32(3.0)
                            + np.float32(0.05)).astype(np.float32)
    for fns, props, final, kind in (([0], [64], 16, "uniform"), ([0, 0], [64, 64], 16, "uniform"), ([0], [64], 16, "lindisp"),
                                     ([0], [33], 7, "lindisp")):
        est = na.PropNetEstimator().to(dev)
        ts, te = est.sampling([fn_t] * len(fns), props, final, n_rays, 2.0, 6.0, sampling_type=kind, requires_grad=True)
        ots, ote, levels = oracle.propnet_sampling([fn_np] * len(fns), props, final, n_rays, 2.0, 6.0, sampling_type=kind)
        assert ts.shape == (n_rays, final)
        scale = 6.0
        assert np.abs(ts.cpu().numpy() - ots).max() <= 1e-5 * scale and np.abs(te.cpu().numpy() - ote).max() <= 1e-5 * scale, kind
        assert len(est.prop_cache) == len(props) + 1
        for (iv, cdfs), (o_iv, o_cdfs) in zip(est.prop_cache[:-1], levels):
            cdfs = cdfs.materialize() if hasattr(cdfs, "materialize") else cdfs   # (kept as the transmittance on the device)
            assert np.abs(iv.vals.cpu().numpy() - o_iv).max() <= 1e-5
            assert np.abs(cdfs.detach().cpu().numpy() - o_cdfs).max() <= 1e-5
        assert est.prop_cache[-1][1] is None
    # and the reference's own outputs (fixture written by oracle/gen_golden.py from the reference's level loop)
    g = load_golden("propnet")
    offg = torch.from_numpy(g["off"]).to(dev)
    fn_g = lambda ts, te: torch.exp(-((ts + te) * 0.5 - 4.0 - offg) ** 2 * 2.0) * 3.0 + 0.05
    for tag, kind in (("u", "uniform"), ("l", "lindisp")):
        props = [int(v) for v in g[f"{tag}_props"]]
        ts, te = na.PropNetEstimator().to(dev).sampling([fn_g] * len(props), props, int(g[f"{tag}_final"]), offg.shape[0], 2.0, 6.0,
                                                        sampling_type=kind)
        assert np.abs(ts.cpu().numpy() - g[f"{tag}_t_starts"]).max() <= 6e-5 and np.abs(te.cpu().numpy() - g[f"{tag}_t_ends"]).max() <= 6e-5


def test_bench_geometry_bit_exact_vs_oracle(dev, oracle):
    """The bench's geometry at 64x64 rays (pinhole camera 2.2 units from the box, 128^3 shell grid, step
    2 sqrt(3)/1024): every ray marches ~650 steps from the near plane before it meets the grid, the case the
    approach table and the Stepper's jumps exist for.  Sampler output and the API's intervals, bit for bit."""
    import bench
    b = bench.make_grid(128, "shell10")
    o, d = bench.make_rays(64 * 64, "image", rank=3)
    step = 2 * 3 ** 0.5 / 1024
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=128).to(dev)
    est.binaries = T(b, dev)
    ab = est.aabbs.cpu().numpy()
    for near in (0.0, 0.37):
        ri, ts, te = est.sampling(T(o, dev), T(d, dev), near_plane=near, render_step_size=step)
        ori, ots, ote = oracle.occgrid_sampling(o, d, b, ab, near_plane=near, render_step_size=step)
        assert ri.numel() > 50000
        assert (ri.cpu().numpy() == ori).all() and (ts.cpu().numpy() == ots).all() and (te.cpu().numpy() == ote).all()
    res = na.traverse_grids(T(o, dev), T(d, dev), T(b, dev), T(ab, dev), step_size=step)
    _cmp_traversal(res, oracle.traverse_grids(o, d, b, ab, step_size=step))


def test_prefetched_traversal_is_the_same_sampling(dev):
    """sampling(traversal=handle) == sampling(): the handle holds the traversal made on the side stream by the same
    kernels; a handle made for other rays is rejected."""
    import bench
    b = bench.make_grid(64, "shell10")
    o, d = bench.make_rays(48 * 48, "image", rank=1)
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=64).to(dev)
    est.binaries = T(b, dev)
    est.occs = T(b.reshape(-1).astype(np.float32), dev)
    ro, rd = T(o, dev), T(d, dev)
    sig = lambda ts, te, ri: 6.0 + 3.0 * torch.sin(9.0 * (ts + te))
    kw = dict(sigma_fn=sig, render_step_size=0.004, early_stop_eps=1e-2, near_plane=0.1)
    ref = est.sampling(ro, rd, **kw)
    for _ in range(3):
        h = est.prefetch_traversal(ro, rd, near_plane=0.1, render_step_size=0.004)
        got = est.sampling(ro, rd, traversal=h, **kw)
        assert all(torch.equal(a, b_) for a, b_ in zip(ref, got)) and ref[0].numel() > 10000
    h = est.prefetch_traversal(ro, rd, near_plane=0.1, render_step_size=0.004)
    with pytest.raises(ValueError):
        est.sampling(ro, rd, traversal=h, **dict(kw, near_plane=0.2))
    with pytest.raises(AssertionError):
        got = est.sampling(ro, rd, traversal=h, **kw); est.sampling(ro, rd, traversal=h, **kw)
    # the double-buffer pattern: rays rewritten IN PLACE after the prefetch must not be served the old traversal; nor may
    # per-ray planes or the stratified flag differ between prefetch and sampling
    h = est.prefetch_traversal(ro, rd, near_plane=0.1, render_step_size=0.004)
    torch.cuda.synchronize()
    ro.add_(0.01)
    with pytest.raises(ValueError):
        est.sampling(ro, rd, traversal=h, **kw)
    h = est.prefetch_traversal(ro, rd, near_plane=0.1, render_step_size=0.004)
    with pytest.raises(ValueError):
        est.sampling(ro, rd, traversal=h, t_min=torch.full((ro.shape[0],), 0.3, device=dev), **kw)
    h = est.prefetch_traversal(ro, rd, near_plane=0.1, render_step_size=0.004)
    with pytest.raises(ValueError):
        est.sampling(ro, rd, traversal=h, stratified=True, **kw)
    ref = est.sampling(ro, rd, **kw)
    # with a cone angle (run records + recurrence expansion, overflow rays on a nested side stream)
    est.binaries = T(np.indices((64, 64, 64)).sum(0)[None] % 2 == 0, dev)       # checkerboard: rays with > 32 chains
    kwc = dict(kw, cone_angle=0.01)
    ref = est.sampling(ro, rd, **kwc)
    h = est.prefetch_traversal(ro, rd, near_plane=0.1, render_step_size=0.004, cone_angle=0.01)
    got = est.sampling(ro, rd, traversal=h, **kwc)
    assert all(torch.equal(a, b_) for a, b_ in zip(ref, got)) and ref[0].numel() > 10000


def test_traversal_fuzz_bit_exact(dev, oracle):
    """Randomised configurations against the oracle, bit for bit: resolutions (also not multiples of 4), 1-3
    levels, occupancies from 2 % to 90 % (90 % random cells = dozens of runs per ray: the overflow path), steps
    from a fraction of a cell to several cells, rays from inside and outside with zero direction components,
    per-ray near / far planes, sample budgets; API traverse_grids (intervals + samples), the sampler's direct
    path and the test-mode (mask + limit) path."""
    rng = np.random.default_rng(int(os.environ.get("NFA_FUZZ_SEED", "2024")))   # NFA_FUZZ_SEED: soak runs with other seeds
    n_cases = int(os.environ.get("NFA_FUZZ_CASES", "24"))
    for case in range(n_cases):
        res = [int(rng.choice([8, 16, 24, 30, 50])) for _ in range(3)] if case % 3 else [int(rng.choice([16, 32]))] * 3
        levels = int(rng.integers(1, 4))
        occ = float(rng.choice([0.02, 0.1, 0.5, 0.9]))
        n_rays = int(rng.integers(200, 1500))
        b = rng.random((levels, *res)) < occ
        if case % 5 == 0:
            b[:] = False
            b[:, res[0] // 4: res[0] // 2] = True               # a slab: long continuous runs
        if case % 7 == 3:                                         # checkerboard: > 32 runs per ray (overflow rays)
            res = [48, 48, 48]
            ii = np.indices(res).sum(0)
            b = np.broadcast_to((ii % 2 == 0), (levels, *res)).copy()
        o = (rng.random((n_rays, 3)) * 3 - 1.5).astype(np.float32)
        if case % 2:
            o *= 0.3                                              # inside the level-0 box
        d = rng.standard_normal((n_rays, 3)).astype(np.float32)
        d[rng.random(n_rays) < 0.1, int(rng.integers(0, 3))] = 0.0    # axis-parallel components
        d /= np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-6)
        step = float(rng.choice([2e-3, 7e-3, 0.03, 0.11])) * (2.0 / min(res)) * 8
        est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=levels).to(dev)
        ab = est.aabbs.cpu().numpy()
        near = (rng.random(n_rays) * 0.5).astype(np.float32) if case % 4 == 1 else None
        far = (near + 0.5 + rng.random(n_rays) * 3).astype(np.float32) if near is not None else None
        kw = dict(step_size=step)
        if near is not None:
            kw.update(near_planes=near, far_planes=far)
        if case % 6 == 2:
            kw.update(traverse_steps_limit=int(rng.integers(3, 40)))
        tkw = {k: (T(v, dev) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
        res_api = na.traverse_grids(T(o, dev), T(d, dev), T(b, dev), T(ab, dev), **tkw)
        ref = oracle.traverse_grids(o, d, b, ab, **kw)
        _cmp_traversal(res_api, ref)
        # sampler path (t_starts / t_ends of the same samples)
        nearp = T(near if near is not None else np.zeros(n_rays, np.float32), dev)
        farp = T(far if far is not None else np.full(n_rays, 1e10, np.float32), dev)
        mask = (rng.random(n_rays) < 0.7) if case % 6 == 2 else None
        ri, ts, te, pi = na.grid._traverse_samples(
            T(o, dev), T(d, dev), T(b, dev), T(ab, dev), nearp, farp, step, 0.0,
            rays_mask=None if mask is None else T(mask, dev), traverse_steps_limit=kw.get("traverse_steps_limit"),
            near_hint=0.0 if near is None else None)
        if mask is None:
            riv, rsm, _ = oracle.traverse_grids(o, d, b, ab, near_planes=nearp.cpu().numpy(), far_planes=farp.cpu().numpy(),
                                                **{k: v for k, v in kw.items() if k not in ("near_planes", "far_planes")})
            L, Rr = riv["vals"][riv["is_left"]], riv["vals"][riv["is_right"]]
            assert (ri.cpu().numpy() == rsm["ray_indices"]).all() and (ts.cpu().numpy() == L).all() and (te.cpu().numpy() == Rr).all(), case
        else:
            riv, rsm, _ = oracle.traverse_grids(o, d, b, ab, near_planes=nearp.cpu().numpy(), far_planes=farp.cpu().numpy(),
                                                step_size=step, traverse_steps_limit=kw["traverse_steps_limit"],
                                                over_allocate=True, rays_mask=mask)
            L, Rr = riv["vals"][riv["is_left"]], riv["vals"][riv["is_right"]]
            keep = rsm["is_valid"]
            assert (ri.cpu().numpy() == rsm["ray_indices"][keep]).all() and (ts.cpu().numpy() == L).all() and (te.cpu().numpy() == Rr).all(), case


def test_rows_dense_region_many_tiny_rays(dev):
    """Thousands of rays of 0, 1 or 2 samples (an image region where almost nothing is hit): far more rays than elements
    per 1024-element range, so tiles end on the 256-ray bound.  `rendering` forward and backward against the closed form
    for such rays, and bit-identical to the same batch without its empty rays."""
    rng = np.random.default_rng(202)
    R = 60_000
    cnt = rng.choice([0, 0, 0, 1, 1, 2], R).astype(np.int64)
    cnt[1000:9000] = 0                                   # a long gap inside
    n = int(cnt.sum())
    ray = np.repeat(np.arange(R), cnt)
    first = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    k = np.arange(n) - first[ray]                        # 0 or 1: position inside the ray
    ts = rng.uniform(0.5, 1.0, n).astype(np.float32) + k.astype(np.float32); te = ts + np.float32(0.25)
    sg = rng.uniform(0.0, 8.0, n).astype(np.float32); rgb = rng.random((n, 3)).astype(np.float32)
    gw = rng.random((R, 3)).astype(np.float32)
    def run(ri_np, n_rays, g_np):
        s = T(sg, dev).requires_grad_(True); c = T(rgb, dev).requires_grad_(True)
        out = na.rendering(T(ts, dev), T(te, dev), torch.from_numpy(ri_np).to(dev), n_rays=n_rays,
                           rgb_sigma_fn=lambda a, b, r: (c, s))
        (out[0] * T(g_np, dev)).sum().backward()
        return out, s.grad, c.grad
    full, gs, gc = run(ray, R, gw)
    # closed form (float64): alpha_k = 1 - exp(-sigma_k delta_k), T_0 = 1, T_1 = exp(-sigma_0 delta_0)
    d64 = (te - ts).astype(np.float64); a64 = 1 - np.exp(-sg.astype(np.float64) * d64)
    prev = np.where(k > 0, np.roll(sg.astype(np.float64) * d64, 1), 0.0)
    w64 = np.exp(-prev) * a64
    col = np.zeros((R, 3)); np.add.at(col, ray, w64[:, None] * rgb.astype(np.float64))
    assert np.abs(full[0].detach().cpu().numpy() - col).max() < 2e-6
    assert np.abs(full[3]["weights"].detach().cpu().numpy() - w64).max() < 2e-6
    assert np.abs(gc.cpu().numpy() - w64[:, None] * gw[ray].astype(np.float64)).max() < 2e-6
    has = cnt > 0
    rows = np.nonzero(has)[0]
    comp, gs_c, gc_c = run(np.repeat(np.arange(rows.size), cnt[rows]), rows.size, gw[rows])
    assert torch.equal(full[0][torch.from_numpy(rows).to(dev)], comp[0]) and bool((full[0][torch.from_numpy(~has).to(dev)] == 0).all())
    assert torch.equal(gs, gs_c) and torch.equal(gc, gc_c)


def test_compact_samples_consecutive_and_arbitrary_output_offsets(dev):
    """nfa_compact_samples through the C ABI: with the running-sum offsets the sampler passes (kept samples of a step are
    consecutive outputs: packed in LDS, written as vectors) and with per-ray output blocks in REVERSE ray order (not
    consecutive: the element-wise fallback) -- every kept sample lands at out_starts[ray] + its rank in the ray."""
    from nerfacc_amd import _backend as B
    from nerfacc_amd._segments import seginfo_from_packed
    rng = np.random.default_rng(77)
    for R, lam, p_keep in ((5000, 30, 0.6), (300, 400, 0.97), (2000, 3, 0.2), (64, 2000, 0.5)):
        cnt = rng.poisson(lam, R).astype(np.int64)
        cnt[rng.random(R) < 0.1] = 0
        n = int(cnt.sum())
        pi = np.stack([np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt], -1)
        vis = (rng.random(n) < p_keep).astype(np.uint8)
        ts = rng.random(n).astype(np.float32); te = ts + 1
        ray = np.repeat(np.arange(R), cnt)
        kept = np.bincount(ray[vis != 0], minlength=R).astype(np.int64)
        m = int(kept.sum())
        seg = seginfo_from_packed(torch.from_numpy(pi).to(dev), n)
        fwd = np.concatenate([[0], np.cumsum(kept)[:-1]])
        rev = np.concatenate([[0], np.cumsum(kept[::-1])[:-1]])[::-1].copy()     # ray R-1 first
        for starts in (fwd, rev):
            o_ri = torch.full((m,), -1, dtype=torch.int64, device=dev)
            o_ts = torch.full((m,), -1.0, device=dev); o_te = torch.full((m,), -1.0, device=dev)
            d = [torch.from_numpy(x).to(dev) for x in (vis, ts, te, starts)]
            B.call("nfa_compact_samples", B.ptr(d[0]), B.ptr(d[1]), B.ptr(d[2]), B.ptr(seg.packed_info), B.ptr(seg.tiles),
                   seg.n_tiles, B.ptr(d[3]), R, n, B.ptr(o_ri), B.ptr(o_ts), B.ptr(o_te), m, B.stream())
            k = vis != 0
            rank = np.concatenate([np.arange(c) for c in kept]) if m else np.zeros(0, np.int64)
            want_pos = starts[ray[k]] + rank
            e_ri = np.full(m, -1, np.int64); e_ts = np.full(m, -1, np.float32); e_te = np.full(m, -1, np.float32)
            e_ri[want_pos] = ray[k]; e_ts[want_pos] = ts[k]; e_te[want_pos] = te[k]
            assert np.array_equal(o_ri.cpu().numpy(), e_ri) and np.array_equal(o_ts.cpu().numpy(), e_ts)
            assert np.array_equal(o_te.cpu().numpy(), e_te)
            # a capacity below the total (outputs sized before the total was known): nothing at or beyond it is written
            cap = max(m * 2 // 3, 1)
            o_ri.fill_(-1); o_ts.fill_(-1.0); o_te.fill_(-1.0)
            B.call("nfa_compact_samples", B.ptr(d[0]), B.ptr(d[1]), B.ptr(d[2]), B.ptr(seg.packed_info), B.ptr(seg.tiles),
                   seg.n_tiles, B.ptr(d[3]), R, n, B.ptr(o_ri), B.ptr(o_ts), B.ptr(o_te), cap, B.stream())
            assert np.array_equal(o_ri.cpu().numpy()[:cap], e_ri[:cap]) and np.array_equal(o_te.cpu().numpy()[:cap], e_te[:cap])
            assert (o_ri[cap:] == -1).all() and (o_ts[cap:] == -1.0).all() and (o_te[cap:] == -1.0).all()


def test_cdf_rows_fused_with_the_transmittance_pass(dev):
    """PropNetEstimator's level step `1 - cat([T(sigma), 0])` as one engine pass (nfa_density_cdf_rows_fwd / _bwd) against
    the two-step form (render_transmittance_from_density, then the complement): same bits forward and backward; ragged
    row lengths around the engine's step / tile sizes."""
    from nerfacc_amd.estimators import prop_net as PN
    rng = np.random.default_rng(31)
    for R, S in ((513, 64), (1000, 17), (3, 1), (257, 300), (70, 1025)):
        ts0 = np.sort(rng.uniform(0.1, 5.0, (R, S + 1)).astype(np.float32), -1)
        ts, te = T(ts0[:, :-1].copy(), dev), T(ts0[:, 1:].copy(), dev)
        sg_np = rng.uniform(0.0, 6.0, (R, S)).astype(np.float32)
        gc = T(rng.normal(size=(R, S + 1)).astype(np.float32), dev)
        outs = []
        for fuse in (True, False):
            PN.FUSE_CDFS = fuse
            try:
                sg = T(sg_np, dev).requires_grad_(True)
                cd = PN._cdfs_from_density(ts, te, sg)
                (cd * gc).sum().backward()
                with torch.no_grad():
                    cd_ng = PN._cdfs_from_density(ts, te, sg.detach())
            finally:
                PN.FUSE_CDFS = True
            outs.append((cd.detach(), sg.grad.clone(), cd_ng))
        (c1, g1, n1), (c0, g0, n0) = outs
        assert c1.shape == (R, S + 1) and torch.equal(c1, c0) and torch.equal(n1, c1) and torch.equal(n0, c0)
        assert torch.equal(g1, g0)
        assert float(c1[:, -1].min()) == 1.0 and float(c1[:, 0].max()) == 0.0      # T_0 = 1, the appended 0


def test_pdf_loss_fused_matches_composition(dev):
    """The batched interlevel loss as one native pass each way against the reference's composition (searchsorted +
    gathers + elementwise, prop_net.py:232-256) evaluated with torch autograd: values and both gradients."""
    from nerfacc_amd.data_specs import RayIntervals
    from nerfacc_amd.estimators.prop_net import _pdf_loss
    from nerfacc_amd.pdf import searchsorted
    rng = np.random.default_rng(5)
    for R, Q1, K1 in ((513, 17, 65), (40, 65, 65), (7, 3, 2), (300, 129, 33)):
        def mk(n):
            v = np.sort(rng.uniform(0, 1, (R, n)).astype(np.float32), -1)
            c = np.sort(rng.uniform(0, 1, (R, n)).astype(np.float32), -1)
            return torch.from_numpy(v).to(dev), torch.from_numpy(c).to(dev)
        qv, qc0 = mk(Q1); kv, kc0 = mk(K1)
        g = torch.from_numpy(rng.normal(size=(R, Q1 - 1)).astype(np.float32)).to(dev)
        qc, kc = qc0.clone().requires_grad_(True), kc0.clone().requires_grad_(True)
        loss = _pdf_loss(RayIntervals(vals=qv), qc, RayIntervals(vals=kv), kc)
        (loss * g).sum().backward()
        # the composition
        qc2, kc2 = qc0.clone().requires_grad_(True), kc0.clone().requires_grad_(True)
        il, ir = searchsorted(RayIntervals(vals=kv), RayIntervals(vals=qv))
        w = qc2[..., 1:] - qc2[..., :-1]
        wo = kc2.gather(-1, ir[..., 1:]) - kc2.gather(-1, il[..., :-1])
        ref = torch.clip(w - wo, min=0) ** 2 / (w + 1e-7)
        (ref * g).sum().backward()
        assert_close(loss, ref, atol=1e-6, rtol=1e-5)
        assert_close(kc.grad, kc2.grad, atol=2e-5, rtol=1e-4)
        assert_close(qc.grad, qc2.grad, atol=2e-4, rtol=1e-3)   # d/dw has a 1/(w+eps)^2 term: large values
    # key gradient only (the estimator detaches the query CDF)
    kc = kc0.clone().requires_grad_(True)
    _pdf_loss(RayIntervals(vals=qv), qc0, RayIntervals(vals=kv), kc).sum().backward()
    assert kc.grad is not None and torch.isfinite(kc.grad).all()


def test_pdf_loss_mean_form_matches_the_loss_array(dev):
    """compute_loss takes the MEAN of the interlevel loss (ref prop_net.py:151): the form that leaves per-wave partial sums and
    takes the mean's scalar gradient equals `_pdf_loss(...).mean()` in value (fp32 summation order aside) and in both gradients,
    for the short-row kernel, the general kernel, a scaled loss (loss_scaler) and the key-gradient-only case of the estimator."""
    from nerfacc_amd.data_specs import RayIntervals
    from nerfacc_amd.estimators.prop_net import _pdf_loss, _pdf_loss_mean
    rng = np.random.default_rng(15)
    for R, Q1, K1 in ((100_003, 17, 65), (513, 17, 65), (40, 65, 65), (7, 3, 2), (300, 129, 33), (1, 2, 1)):
        def mk(n):
            v = np.sort(rng.uniform(0, 1, (R, n)).astype(np.float32), -1)
            c = np.sort(rng.uniform(0, 1, (R, n)).astype(np.float32), -1)
            return torch.from_numpy(v).to(dev), torch.from_numpy(c).to(dev)
        qv, qc0 = mk(Q1); kv, kc0 = mk(K1)
        qc, kc = qc0.clone().requires_grad_(True), kc0.clone().requires_grad_(True)
        m = _pdf_loss_mean(RayIntervals(vals=qv), qc, RayIntervals(vals=kv), kc)
        assert m.dim() == 0
        (m * 3.0).backward()
        qc2, kc2 = qc0.clone().requires_grad_(True), kc0.clone().requires_grad_(True)
        ref = _pdf_loss(RayIntervals(vals=qv), qc2, RayIntervals(vals=kv), kc2).mean()
        (ref * 3.0).backward()
        assert_close(m, ref, atol=1e-7, rtol=2e-5)
        assert_close(kc.grad, kc2.grad, atol=1e-9, rtol=1e-5)
        assert_close(qc.grad, qc2.grad, atol=1e-8, rtol=1e-5)
        # twice the same launch: the same bits (per-wave partial sums, no atomics)
        m2 = _pdf_loss_mean(RayIntervals(vals=qv), qc0, RayIntervals(vals=kv), kc0)
        assert torch.equal(m2, m.detach())
    kc = kc0.clone().requires_grad_(True)
    _pdf_loss_mean(RayIntervals(vals=qv), qc0, RayIntervals(vals=kv), kc).backward()
    assert kc.grad is not None and torch.isfinite(kc.grad).all()


# ----------------------------------------------------------------------------- full-size properties (BASELINE cfg 2)
def test_full_size_properties(dev):
    """1024x1024 rays through a 128^3 grid at ~10% occupancy: size-independent invariants."""
    import bench
    w = bench.make_workload(dev, n_rays=1024 * 1024, res=128)
    est = w["estimator"]
    ri, ts, te = est.sampling(w["rays_o"], w["rays_d"], sigma_fn=w["sigma_fn"], render_step_size=w["step"],
                              early_stop_eps=1e-4, alpha_thre=0.0)
    n = w["rays_o"].shape[0]
    assert ri.dtype == torch.int64 and (ri[1:] >= ri[:-1]).all() and ri.min() >= 0 and ri.max() < n
    assert (te > ts).all() and (ts >= 0).all()
    pi = na.pack_info(ri, n)
    assert int(pi[:, 1].sum()) == ri.numel() and (pi[1:, 0] == pi[:-1, 0] + pi[:-1, 1]).all()
    assert int(pi[:, 1].max()) <= 1024
    # same ray: samples are ordered and disjoint
    same = ri[1:] == ri[:-1]
    assert (ts[1:][same] >= te[:-1][same] - 1e-6).all()
    sig = w["sigma_fn"](ts, te, ri)
    wts, tr, al = na.render_weight_from_density(ts, te, sig, ray_indices=ri, n_rays=n)
    opac = na.accumulate_along_rays(wts, None, ri, n)
    assert (opac <= 1.0 + 1e-4).all() and (wts >= 0).all() and (tr <= 1.0).all()
    # telescoping identity: sum_k w_k = 1 - T_last * (1 - alpha_last)
    last = torch.zeros(n, dtype=torch.bool, device=dev); has = pi[:, 1] > 0
    idx_last = (pi[:, 0] + pi[:, 1] - 1)[has]
    resid = 1.0 - tr[idx_last] * (1.0 - al[idx_last])
    assert torch.allclose(opac[has, 0], resid, atol=2e-5)
    # linearity of the accumulation
    v = torch.rand((ri.numel(), 3), device=dev)
    a1 = na.accumulate_along_rays(wts, v, ri, n); a2 = na.accumulate_along_rays(wts, 2 * v, ri, n)
    assert torch.allclose(a2, 2 * a1, atol=1e-5)
    # checksum against torch's atomics path
    ref = torch.zeros((n, 3), device=dev).index_add_(0, ri, wts[:, None] * v)
    assert torch.allclose(a1, ref, atol=2e-5)
    del last


# ----------------------------------------------------------------------------- the reference's `_C` surface
def test_cuda_compat_module_matches_pybind_surface(dev, oracle):
    """nerfacc_amd.cuda_compat exposes the names/signatures of nerfacc/cuda/csrc/nerfacc.cpp:100-129; the calls
    below are written the way the reference's Python layer makes them (scan.py:197, grid.py:165, pdf.py:59,125)."""
    import nerfacc_amd.cuda_compat as _C
    g = load_golden("ragged_packed")
    starts, cnts = T(g["packed_info"][:, 0].copy(), dev), T(g["packed_info"][:, 1].copy(), dev)
    x, xp, gg = T(g["x"], dev), T(g["xp"], dev), T(g["g"], dev)
    assert_close(_C.inclusive_sum(starts, cnts, x, False, False), g["inclusive_sum"], atol=1e-5)
    assert_close(_C.exclusive_sum(starts, cnts, gg, False, True), g["exclusive_sum_grad"], atol=2e-5, rtol=1e-4)
    out = _C.exclusive_prod_forward(starts, cnts, xp)
    assert_close(out, g["exclusive_prod"], atol=1e-6)
    assert_close(_C.exclusive_prod_backward(starts, cnts, xp, out, gg), g["exclusive_prod_grad"], atol=2e-5, rtol=1e-4)
    out = _C.inclusive_prod_forward(starts, cnts, xp)
    assert_close(_C.inclusive_prod_backward(starts, cnts, xp, out, gg), g["inclusive_prod_grad"], atol=2e-5, rtol=1e-4)
    with pytest.raises(RuntimeError):
        _C.inclusive_sum(starts.cpu(), cnts, x, False, False)
    # grid: the 17-argument traverse_grids of grid.py:165-185
    gt = load_golden("traversal")
    binaries = np.unpackbits(gt["a_binaries"]).astype(bool).reshape(4, 32, 32, 32)
    o, d, ab = gt["a_rays_o"][:8], gt["a_rays_d"][:8], gt["a_aabbs"]
    to, td, tab, tb = T(o, dev), T(d, dev), T(ab, dev), T(binaries, dev)
    t_mins, t_maxs, hits = _C.ray_aabb_intersect(to, td, tab, -float("inf"), float("inf"), float("inf"))
    t_sorted, t_indices = torch.sort(torch.cat([t_mins, t_maxs], -1), -1)
    iv, sm, term = _C.traverse_grids(to, td, torch.ones(8, dtype=torch.bool, device=dev), tb, tab, t_sorted, t_indices, hits,
                                     torch.zeros(8, device=dev), torch.full((8,), float("inf"), device=dev), 1e-3, 0.0,
                                     True, True, True, -1, False)
    assert (iv.vals.cpu().numpy() == gt["a8_iv_vals"]).all() and (iv.is_left.cpu().numpy() == gt["a8_iv_left"]).all()
    assert (torch.stack([sm.chunk_starts, sm.chunk_cnts], -1).cpu().numpy() == gt["a8_sm_packed"]).all()
    assert (term.cpu().numpy() == gt["a8_term"]).all() and sm.is_left is None and iv.is_valid is None
    # pdf
    gp = load_golden("pdf")
    spec = _C.RaySegmentsSpec(); spec.vals = T(gp["b_vals"], dev)
    ivs, sms = _C.importance_sampling(spec, T(gp["b_cdfs"], dev), int(gp["b_S"]), False)
    assert_close(ivs.vals, gp["b_oracle_edges"], atol=1e-6); assert_close(sms.vals, gp["b_oracle_centres"], atol=1e-6)
    q, k = _C.RaySegmentsSpec(), _C.RaySegmentsSpec()
    q.vals, k.vals = T(gp["loss_q_vals"], dev), T(gp["loss_k_vals"], dev)
    il, ir = _C.searchsorted(q, k)
    assert (il.cpu().numpy() == gp["loss_ids_left"]).all() and (ir.cpu().numpy() == gp["loss_ids_right"]).all()


# ----------------------------------------------------------------------------- test-mode marching loop
@pytest.mark.parametrize("levels,cone,alpha_thre", [(1, 0.0, 0.0), (2, 0.0, 0.02), (2, 0.004, 0.0)])
def test_test_mode_marching_loop(dev, oracle, levels, cone, alpha_thre):
    from nerfacc_amd.marching import render_rays_test_mode
    rng = np.random.default_rng(17)
    n, res, step = 1500, 48, 6e-3
    o = (rng.random((n, 3)).astype(np.float32) - 0.5) * 3.0
    d = rng.standard_normal((n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    b = rng.random((levels, res, res, res)) < 0.25
    est = na.OccGridEstimator([-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=levels).to(dev)
    est.binaries = T(b, dev)
    bk = np.array([0.2, 0.4, 0.6], np.float32)

    def field_np(ts, te, ri):
        tm = (ts + te) * np.float32(0.5)
        sig = (np.float32(25.0) * (np.float32(0.5) + np.float32(0.5) * np.sin(np.float32(9.0) * tm))).astype(np.float32)
        rgbs = np.stack([np.float32(0.5) + np.float32(0.5) * np.cos(tm), (ri % 7).astype(np.float32) / np.float32(7.0),
                         np.full_like(tm, 0.3)], -1).astype(np.float32)
        return rgbs, sig

    def field_t(ts, te, ri):
        tm = (ts + te) * 0.5
        return (torch.stack([0.5 + 0.5 * torch.cos(tm), (ri % 7).float() / 7.0, torch.full_like(tm, 0.3)], -1),
                25.0 * (0.5 + 0.5 * torch.sin(9.0 * tm)))

    rgb, opa, dep, total = render_rays_test_mode(600, field_t, est, T(o, dev), T(d, dev), near_plane=0.05, far_plane=1e10,
                                                 render_step_size=step, render_bkgd=T(bk, dev), cone_angle=cone,
                                                 alpha_thre=alpha_thre, early_stop_eps=1e-3)
    orgb, oopa, odep, ototal, info = oracle.test_mode_loop(600, field_np, o, d, b, est.aabbs.cpu().numpy(), 0.05, 1e10, step, bk,
                                                           cone, alpha_thre, 1e-3, guard=2e-6)
    assert ototal > 10000
    # Rays that end an iteration within 2e-6 of the early-termination threshold (opacities agree to ~1e-7; near the threshold
    # a sample moves the opacity by ~1e-4, so a few percent of the rays pass that close) may live one iteration longer on
    # one side: at most 64 samples each.  Every other ray is identical, and without such rays so is the sample count.
    g = info["guard_rays"]
    assert g.mean() < 0.1
    assert abs(total - ototal) <= 64 * int(g.sum())
    k = torch.from_numpy(~g).to(dev)
    assert_close(opa[k], oopa[~g], atol=2e-5, rtol=1e-5); assert_close(rgb[k], orgb[~g], atol=2e-5, rtol=1e-5)
    assert_close(dep[k], odep[~g], atol=1e-4, rtol=1e-4)
    assert (opa.max() <= 1.0 + 1e-5) and (opa.min() >= 0)


def _cfg5_scene(dev, res=512, levels=4, seed=5):
    """BASELINE cfg 5's scene (bench.extra_cfg5): nested levels, a shell r in (0.5, 0.66) of each level's own box + 2 % speckle."""
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=levels).to(dev)
    ax = (torch.arange(res, device=dev, dtype=torch.float32) + 0.5) / res * 2 - 1
    r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
    shell = (r > 0.5) & (r < 0.66)
    g = torch.Generator(device=dev); g.manual_seed(seed)
    b = torch.stack([shell | (torch.rand((res, res, res), device=dev, generator=g) < 0.02) for _ in range(levels)])
    est.binaries = b
    est.occs = b.reshape(-1).float()
    return est


def _cfg5_field_np(ts, te, ri):
    tm = (ts + te) * np.float32(0.5)
    sig = (np.float32(6.0) * (np.float32(0.5) + np.float32(0.5) * np.sin(np.float32(11.0) * tm))).astype(np.float32)
    rgbs = np.stack([np.float32(0.5) + np.float32(0.5) * np.cos(tm), (ri % 5).astype(np.float32) / np.float32(5.0),
                     np.full_like(tm, 0.25)], -1).astype(np.float32)
    return rgbs, sig


def _cfg5_field_t(ts, te, ri):
    tm = (ts + te) * 0.5
    return (torch.stack([0.5 + 0.5 * torch.cos(tm), (ri % 5).float() / 5.0, torch.full_like(tm, 0.25)], -1),
            6.0 * (0.5 + 0.5 * torch.sin(11.0 * tm)))


def test_cfg5_regime_sampling_bit_exact(dev, oracle):
    """BASELINE cfg 5 at its real grid: 4 nested 512^3 levels (64 MiB bit copy, the walk's resolution limit, the brick grid of
    the cone kernels), rays from inside the level-0 box, step 1e-3, cone 0.004, near 0.2, alpha_thre 1e-2, early_stop_eps
    1e-4 (ref: examples/train_ngp_nerf_occ.py:64-75) -- OccGridEstimator.sampling against the oracle bit for bit (guard
    band on the visibility thresholds as everywhere), with the cone angle (run records of the count pass + recurrence
    expansion) and without it (run-length walk through 4 levels)."""
    from oracle import check as OC
    est = _cfg5_scene(dev)
    b = est.binaries.cpu().numpy()
    ab = est.aabbs.cpu().numpy()
    rng = np.random.default_rng(55)
    R = 3000
    o = (rng.random((R, 3)).astype(np.float32) - 0.5)
    d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    sig_np = lambda ts, te, ri: _cfg5_field_np(ts, te, ri)[1]
    sig_t = lambda ts, te, ri: _cfg5_field_t(ts, te, ri)[1]
    for cone in (0.004, 0.0):
        a_thre = 1e-2 if cone > 0 else 2e-3      # (constant 1e-3 steps: opacities stay below 1e-2 in this field)
        kw = dict(near_plane=0.2, render_step_size=1e-3, cone_angle=cone, alpha_thre=a_thre, early_stop_eps=1e-4)
        ri, ts, te = est.sampling(T(o, dev), T(d, dev), sigma_fn=sig_t, **kw)
        (ori, ots, ote), (fri, fts, fte, fpi) = oracle.occgrid_sampling(o, d, b, ab, sigma_fn=sig_np, occs_mean=float(b.mean()),
                                                                         return_all=True, **kw)
        assert fri.size > 300_000 and 0 < ori.size < fri.size               # the regime: ~190 samples per ray, two thirds of them dropped
        tr, al = oracle.render_transmittance_from_density(fts, fte, sig_np(fts, fte, fri), fpi)
        ok, info = OC.compare_sampling((ri.cpu().numpy(), ts.cpu().numpy(), te.cpu().numpy()), (ori, ots, ote), (fri, fts, fte), tr, al,
                                       early_stop_eps=1e-4, alpha_thre=min(a_thre, float(b.mean())))
        assert ok, (cone, info)
        # the traversal alone: every sample, bit for bit
        ria, tsa, tea = est.sampling(T(o, dev), T(d, dev), near_plane=0.2, render_step_size=1e-3, cone_angle=cone)
        assert (ria.cpu().numpy() == fri).all() and (tsa.cpu().numpy() == fts).all() and (tea.cpu().numpy() == fte).all(), cone


def test_cfg5_regime_test_mode_loop(dev, oracle):
    """SURVEY 8 row a12 at cfg 5's regime (4 nested 512^3 levels, cone 0.004, max_samples 1024): render_rays_test_mode against
    the oracle's restatement of examples/utils.py:252-425 -- the same sample count (no ray of the committed case sits on the
    early-termination threshold) and images within 2e-5."""
    from nerfacc_amd.marching import render_rays_test_mode
    est = _cfg5_scene(dev)
    b = est.binaries.cpu().numpy()
    ab = est.aabbs.cpu().numpy()
    rng = np.random.default_rng(56)
    R = 1200
    o = (rng.random((R, 3)).astype(np.float32) - 0.5)
    d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    bk = np.array([0.1, 0.3, 0.5], np.float32)
    kw = dict(near_plane=0.2, far_plane=1e10, render_step_size=1e-3, cone_angle=0.004, alpha_thre=1e-2, early_stop_eps=1e-4)
    rgb, opa, dep, total = render_rays_test_mode(1024, _cfg5_field_t, est, T(o, dev), T(d, dev), render_bkgd=T(bk, dev), **kw)
    orgb, oopa, odep, ototal, info = oracle.test_mode_loop(1024, _cfg5_field_np, o, d, b, ab, render_bkgd=bk, guard=2e-6, **kw)
    assert ototal > 100_000 and info["iterations"] > 20
    assert not info["guard_rays"].any()
    assert total == ototal
    assert_close(opa, oopa, atol=2e-5, rtol=1e-5); assert_close(rgb, orgb, atol=2e-5, rtol=1e-5)
    assert_close(dep, odep, atol=1e-4, rtol=1e-4)


def test_full_size_bit_exact_vs_oracle(dev, oracle):
    """BASELINE cfg 2 at FULL size (1024x1024 image rays, 128^3 shell10 grid, step 2 sqrt(3)/1024): the sampler's
    (ray_indices, t_starts, t_ends) and packed_info bit for bit against the oracle, colours of rendering() within 1e-5.
    The same comparison runs inside bench.py after the timed loops (`parity_checked`)."""
    import bench
    from oracle import check as OC
    w = bench.make_workload(dev, n_rays=1024 * 1024, res=128)
    n = w["n_rays"]
    bench.run_step(w, 1)
    ri, ts, te, colors = w["last"]
    o, d = w["rays_np"]
    b = w["estimator"].binaries.cpu().numpy()
    aabb = np.array([[-1, -1, -1, 1, 1, 1]], np.float32)
    kept, full, ocolors, _, sig = bench._oracle_step(oracle, o, d, b, aabb, w["step"], 1.0)
    got = (ri.cpu().numpy(), ts.cpu().numpy(), te.cpu().numpy())
    assert got[0].size > 30_000_000
    tr, al = oracle.render_transmittance_from_density(full[1], full[2], sig(full[1], full[2], full[0]), full[3])
    ok, info = OC.compare_sampling(got, kept, full[:3], tr, al, early_stop_eps=1e-4)
    assert ok and info["identical"], info
    assert (na.pack_info(ri, n).cpu().numpy() == oracle.pack_info(kept[0], n)).all()
    assert np.abs(colors.detach().cpu().numpy() - ocolors).max() <= 1e-5 * max(1.0, float(np.abs(ocolors).max()))


def test_cfg4_shared_256_grid(dev, oracle):
    """BASELINE cfg 4's workload on one GPU: the shared 256^3 shell10 grid (built on rank 0 and unpacked from the
    bit-packed broadcast buffer, bench.shared_grid), rank-seeded cameras.  64x64 rays bit-exact against the oracle (sampler
    and the API's interval stream), size-independent properties on the full 1 M-ray batch."""
    import bench
    binaries = bench.shared_grid(dev, 256, "shell10", 0, 1)
    b = bench.make_grid(256, "shell10")
    assert (binaries.cpu().numpy() == b).all()
    step = 2 * 3 ** 0.5 / 1024
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=256).to(dev)
    est.binaries = binaries
    ab = est.aabbs.cpu().numpy()
    for rank in (0, 5):
        o, d = bench.make_rays(64 * 64, "image", rank=rank)
        ri, ts, te = est.sampling(T(o, dev), T(d, dev), render_step_size=step)
        ori, ots, ote = oracle.occgrid_sampling(o, d, b, ab, render_step_size=step)
        assert ri.numel() > 50000
        assert (ri.cpu().numpy() == ori).all() and (ts.cpu().numpy() == ots).all() and (te.cpu().numpy() == ote).all()
    res = na.traverse_grids(T(o, dev), T(d, dev), binaries, T(ab, dev), step_size=step)
    _cmp_traversal(res, oracle.traverse_grids(o, d, b, ab, step_size=step))
    # full size: one step of the bench on this grid
    w = bench.make_workload(dev, n_rays=1024 * 1024, res=256, rank=3, binaries=binaries)
    bench.run_step(w, 1)
    ri, ts, te, colors = w["last"]
    n = w["n_rays"]
    assert (ri[1:] >= ri[:-1]).all() and ri.min() >= 0 and ri.max() < n and (te > ts).all()
    pi = na.pack_info(ri, n)
    assert int(pi[:, 1].sum()) == ri.numel() and int(pi[:, 1].max()) <= 1024
    same = ri[1:] == ri[:-1]
    assert (ts[1:][same] >= te[:-1][same]).all()
    # every sample's mid-point lies in an occupied cell (the reference's own traversal property, tests/test_grid.py:57-68)
    mid = (ts + te) * 0.5
    pos = w["rays_o"][ri] + w["rays_d"][ri] * mid[:, None]
    cell = ((pos + 1.0) * 0.5 * 256).long().clamp(0, 255)
    occ = binaries[0, cell[:, 0], cell[:, 1], cell[:, 2]]
    assert occ.float().mean() > 0.9995   # (mid-points within rounding of a cell face may land in the neighbour)
    assert torch.isfinite(colors).all()


# ----------------------------------------------------------------------------- grid maintenance
def test_grid_update_kernels_vs_oracle(dev, oracle):
    """OccGridEstimator._update on the device (csrc/gridupd.hip) against the oracle's restatement of
    estimators/occ_grid.py:368-404, starting from a state_dict the REFERENCE wrote (fixture): cell positions bit for bit,
    occupancies after a warm-up update and after a sampled update (cells drawn several times: the largest candidate) bit
    for bit, the re-binarised grid (cells within 1e-7 of the threshold excepted) -- and the traversal through the
    bit-packed copy the update leaves behind."""
    g = load_golden("occgrid")
    res = [int(v) for v in g["sd_resolution"]]
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=2).to(dev)
    est.load_state_dict({k: torch.from_numpy(g["sd_" + k]) for k in ("resolution", "aabbs", "occs", "binaries")}, strict=True)
    assert est.occs.is_cuda and torch.equal(est.binaries.cpu(), torch.from_numpy(g["sd_binaries"]))
    ab = est.aabbs.cpu().numpy()
    # (a) cell positions
    rng = np.random.default_rng(3)
    idx = rng.integers(0, est.cells_per_lvl, 5000)
    jit = rng.random((5000, 3)).astype(np.float32)
    x = est._cell_points(1, T(idx, dev), T(jit, dev))
    assert (x.cpu().numpy() == oracle.grid_cell_points(idx, jit, res, ab[1])).all()
    # (b) the reference's own warm-up step: same cells, same positions -> same occupancies
    est.occs = T(g["warm_occs_before"], dev)
    field = lambda p: torch.exp(-4.0 * (p.norm(dim=-1, keepdim=True) - 0.7) ** 2) * 0.05
    field_np = lambda p: (np.exp(np.float32(-4.0) * (np.linalg.norm(p, axis=-1).astype(np.float32) - np.float32(0.7)) ** 2)
                          * np.float32(0.05)).astype(np.float32)
    for step, dup in ((0, False), (300, True)):
        seen_x, seen_idx = [], []
        orig = est._get_all_cells if step == 0 else est._sample_uniform_and_occupied_cells
        def wrapped(*a, _o=orig, **k):
            r = _o(*a, **k); seen_idx.append([t.cpu().numpy() for t in r]); return r
        setattr(est, "_get_all_cells" if step == 0 else "_sample_uniform_and_occupied_cells", wrapped)
        before = est.occs.cpu().numpy().copy()
        def occ_eval(p):
            seen_x.append(p.cpu().numpy()); return field(p)
        est.train()
        est.update_every_n_steps(step, occ_eval, occ_thre=0.02, ema_decay=0.9, warmup_steps=256, n=4)
        setattr(est, "_get_all_cells" if step == 0 else "_sample_uniform_and_occupied_cells", orig)
        exp = before
        for lvl, (ids, px) in enumerate(zip(seen_idx[0], seen_x)):
            occ = field(T(px, dev)).squeeze(-1).cpu().numpy()      # the product's own field values (exp ulps are the field's business)
            assert np.abs(occ - field_np(px)).max() < 1e-7
            cells = lvl * est.cells_per_lvl + ids
            if dup:
                assert len(np.unique(cells)) < len(cells)           # the sampled update really draws cells more than once
            exp = oracle.grid_ema_update(exp, cells, occ, 0.9)
        assert (est.occs.cpu().numpy() == exp).all()
        b, thre = oracle.grid_rebinarize(exp, tuple(est.binaries.shape), 0.02)
        edge = np.abs(exp - thre) < 1e-7
        assert ((est.binaries.cpu().numpy() == b) | edge.reshape(b.shape)).all() and est.binaries.dtype == torch.bool
        # (c) the traversal reads the bit-packed copy the update wrote: same samples as through a freshly packed grid
        assert getattr(est.binaries, "_nfa_walk_bits", None) is not None
        o = (rng.random((700, 3)) * 3 - 1.5).astype(np.float32)
        d = rng.standard_normal((700, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
        got = est.sampling(T(o, dev), T(d, dev), render_step_size=0.01)
        fresh = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=2).to(dev)
        fresh.binaries = est.binaries.clone()
        ref = fresh.sampling(T(o, dev), T(d, dev), render_step_size=0.01)
        assert all(torch.equal(a, b_) for a, b_ in zip(got, ref)) and got[0].numel() > 1000
        ori, ots, ote = oracle.occgrid_sampling(o, d, est.binaries.cpu().numpy(), ab, render_step_size=0.01)
        assert (got[0].cpu().numpy() == ori).all() and (got[1].cpu().numpy() == ots).all()
    # the warm-up result equals what the REFERENCE computed from the same state (its field values differ by exp ulps at most)
    # (checked on the first pass through the fixture's occupancies)


def test_reference_state_dict_and_mark_invisible_cells_gpu(dev, oracle):
    """SURVEY 8 f3 on the device: a reference-written state_dict (fixture) loaded with strict=True, sampled through, and
    mark_invisible_cells (ref occ_grid.py:262-332) against the reference's marking of the same cameras."""
    g = load_golden("occgrid")
    res = [int(v) for v in g["sd_resolution"]]
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=2).to(dev)
    missing = est.load_state_dict({k: torch.from_numpy(g["sd_" + k]) for k in ("resolution", "aabbs", "occs", "binaries")}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    b = g["sd_binaries"]
    rng = np.random.default_rng(11)
    o = (rng.random((900, 3)) * 3 - 1.5).astype(np.float32)
    d = rng.standard_normal((900, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    ri, ts, te = est.sampling(T(o, dev), T(d, dev), render_step_size=0.01)
    ori, ots, ote = oracle.occgrid_sampling(o, d, b, g["sd_aabbs"], render_step_size=0.01)
    assert ri.numel() > 1000 and (ri.cpu().numpy() == ori).all() and (ts.cpu().numpy() == ots).all() and (te.cpu().numpy() == ote).all()
    est.occs.zero_()
    est.mark_invisible_cells(T(g["K"], dev), T(g["c2w"], dev), int(g["W"]), int(g["H"]), near_plane=float(g["near"]))
    assert int((est.occs.cpu().numpy() != g["occs_marked"]).sum()) <= 8
    # round trip: what we save loads back, and the reference's buffer names are the only persistent ones
    sd = est.state_dict()
    assert list(sd.keys()) == ["resolution", "aabbs", "occs", "binaries"]
    est2 = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=2).to(dev)
    est2.load_state_dict(sd, strict=True)
    assert torch.equal(est2.occs, est.occs) and torch.equal(est2.binaries, est.binaries)


# ----------------------------------------------------------------------------- packed resampling / packed loss
def test_importance_sampling_per_ray_counts_packed(dev, oracle):
    """importance_sampling(Tensor n_intervals_per_ray): packed outputs (ref pdf.py:92-105; the reference's host code
    allocates zero samples, pdf.cu:324) against the oracle's per-ray restatement of the reference's two kernels -- batched
    and packed inputs, counts of 0 and 1, masks / ray_indices / packed_info -- and against the int overload when every
    count is the same."""
    rng = np.random.default_rng(21)
    R, E = 300, 33
    v = np.sort(rng.uniform(0, 1, (R, E)).astype(np.float32), -1)
    c = np.sort(rng.uniform(0, 1, (R, E)).astype(np.float32), -1)
    counts = rng.integers(0, 41, R)
    counts[:5] = [0, 1, 2, 40, 1]
    iv, sm = na.importance_sampling(na.RayIntervals(vals=T(v, dev)), T(c, dev), T(counts, dev))
    oiv, osm = oracle.importance_sampling_packed(v, c, counts)
    assert sm.vals.shape == (int(counts.sum()),) and iv.vals.shape == (int(((counts + 1) * (counts > 0)).sum()),)
    assert_close(sm.vals, osm["vals"], atol=1e-6)
    assert_close(iv.vals, oiv["vals"], atol=1e-6)
    assert (sm.ray_indices.cpu().numpy() == osm["ray_indices"]).all() and (iv.ray_indices.cpu().numpy() == oiv["ray_indices"]).all()
    assert (sm.packed_info.cpu().numpy() == osm["packed_info"]).all() and (iv.packed_info.cpu().numpy() == oiv["packed_info"]).all()
    assert (iv.is_left.cpu().numpy() == oiv["is_left"]).all() and (iv.is_right.cpu().numpy() == oiv["is_right"]).all()
    assert iv.is_left.dtype == torch.bool and sm.ray_indices.dtype == torch.int64
    # flattened input segments of different lengths
    lens = rng.integers(2, 50, R)
    pi = np.stack([np.cumsum(lens) - lens, lens], -1)
    fv = np.concatenate([np.sort(rng.uniform(0, 1, n).astype(np.float32)) for n in lens])
    fc = np.concatenate([np.sort(rng.uniform(0, 1, n).astype(np.float32)) for n in lens])
    iv2, sm2 = na.importance_sampling(na.RayIntervals(vals=T(fv, dev), packed_info=T(pi, dev)), T(fc, dev), T(counts, dev))
    oiv2, osm2 = oracle.importance_sampling_packed(fv, fc, counts, packed_info=pi)
    assert_close(sm2.vals, osm2["vals"], atol=1e-6); assert_close(iv2.vals, oiv2["vals"], atol=1e-6)
    assert (iv2.is_left.cpu().numpy() == oiv2["is_left"]).all() and (iv2.packed_info.cpu().numpy() == oiv2["packed_info"]).all()
    # equal counts == the int overload
    same = torch.full((R,), 16, dtype=torch.int64, device=dev)
    iv3, sm3 = na.importance_sampling(na.RayIntervals(vals=T(v, dev)), T(c, dev), same)
    iv4, sm4 = na.importance_sampling(na.RayIntervals(vals=T(v, dev)), T(c, dev), 16)
    assert torch.equal(sm3.vals.view(R, 16), sm4.vals) and torch.equal(iv3.vals.view(R, 17), iv4.vals)
    # one sample per ray, int overload: the ray's whole range (defined here; out of bounds upstream)
    iv5, sm5 = na.importance_sampling(na.RayIntervals(vals=T(v, dev)), T(c, dev), 1)
    assert iv5.vals.shape == (R, 2) and torch.equal(iv5.vals[:, 0], T(v, dev)[:, 0]) and torch.equal(iv5.vals[:, 1], T(v, dev)[:, -1])
    o1iv, o1sm = oracle.importance_sampling_packed(v, c, np.ones(R, np.int64))
    assert_close(sm5.vals.reshape(-1), o1sm["vals"], atol=1e-6)
    # stratified draws stay inside their strata; nothing for all-zero counts
    ivs, sms = na.importance_sampling(na.RayIntervals(vals=T(v, dev)), T(c, dev), T(counts, dev), stratified=True)
    assert sms.vals.shape == sm.vals.shape and torch.isfinite(sms.vals).all()
    iv0, sm0 = na.importance_sampling(na.RayIntervals(vals=T(v, dev)), T(c, dev), torch.zeros(R, dtype=torch.int64, device=dev))
    assert sm0.vals.numel() == 0 and iv0.vals.numel() == 0


def test_pdf_loss_packed_branch(dev, oracle):
    """_pdf_loss with flattened query / key intervals (ref estimators/prop_net.py:244-253, "TODO: not tested" upstream)
    against the oracle's restatement, and against the batched branch on equal-length chunks."""
    from nerfacc_amd.estimators.prop_net import _pdf_loss
    rng = np.random.default_rng(22)
    R = 120
    v = np.sort(rng.uniform(0, 1, (R, 40)).astype(np.float32), -1)
    c = np.sort(rng.uniform(0, 1, (R, 40)).astype(np.float32), -1)
    c[:, 0] = 0; c[:, -1] = 1
    key = na.RayIntervals(vals=T(v, dev))
    q_counts = rng.integers(1, 12, R)
    q_iv, _ = na.importance_sampling(key, T(c, dev), T(q_counts, dev))
    # query CDF per ray: 0 .. 1 over its edges (w sums to one per ray); the key mass is scaled down so that the loss bites
    qpi = q_iv.packed_info.cpu().numpy()
    q_cdfs_np = np.concatenate([np.sort(np.concatenate([[0.0], rng.uniform(0, 1, n - 2), [1.0]])) for n in qpi[:, 1]]).astype(np.float32)
    q_cdfs = T(q_cdfs_np, dev)
    c = (c * np.float32(0.3)).astype(np.float32)
    k_pi = np.stack([np.arange(R) * 40, np.full(R, 40)], -1)
    key_flat = na.RayIntervals(vals=T(v.reshape(-1), dev), packed_info=T(k_pi, dev))
    kc = T(c.reshape(-1), dev).clone().requires_grad_(True)
    loss = _pdf_loss(q_iv, q_cdfs, key_flat, kc)
    ref = oracle.pdf_loss_packed(q_iv.vals.cpu().numpy(), q_cdfs.cpu().numpy(), q_iv.packed_info.cpu().numpy(),
                                 q_iv.is_left.cpu().numpy(), q_iv.is_right.cpu().numpy(), v.reshape(-1), c.reshape(-1), k_pi)
    assert loss.shape == (int(q_counts.sum()),)
    assert_close(loss, ref, atol=1e-6, rtol=1e-5)
    loss.sum().backward()
    assert kc.grad is not None and torch.isfinite(kc.grad).all() and kc.grad.abs().sum() > 0
    # equal counts: the packed branch == the batched (fused) branch
    q2, _ = na.importance_sampling(key, T(c, dev), torch.full((R,), 9, dtype=torch.int64, device=dev))
    qc2 = torch.sort(torch.rand((R, 10), device=dev))[0]
    packed = _pdf_loss(q2, qc2.reshape(-1), key_flat, T(c.reshape(-1), dev))
    batched = _pdf_loss(na.RayIntervals(vals=q2.vals.view(R, 10)), qc2, key, T(c, dev))
    assert_close(packed.view(R, 9), batched, atol=1e-6, rtol=1e-5)


def test_speculative_expansion_capacity(dev, oracle):
    """The sampler launches the expansion into arrays sized from the PREVIOUS batch of the same shape before the host
    has read this batch's total (nerfacc_amd/grid.py): a batch with more samples than that capacity must be expanded
    again into arrays of the right size, a smaller one returns views -- both bit-identical to the oracle."""
    rng = np.random.default_rng(31)
    R, res, step = 3000, 32, 0.01
    o = (rng.random((R, 3)) * 3 - 1.5).astype(np.float32)
    d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    ab = np.array([[-1, -1, -1, 1, 1, 1]], np.float32)
    sizes = []
    for occ in (0.02, 0.6, 0.1, 0.6, 0.0, 0.3):     # small, much larger (over capacity), smaller (views), ..., empty, again
        b = rng.random((1, res, res, res)) < occ
        ri, ts, te, pi = na.grid._traverse_samples(T(o, dev), T(d, dev), T(b, dev), T(ab, dev), torch.zeros(R, device=dev),
                                                   torch.full((R,), 1e10, device=dev), step, 0.0, near_hint=0.0)
        ori, ots, ote = oracle.occgrid_sampling(o, d, b, ab, render_step_size=step)
        assert (ri.cpu().numpy() == ori).all() and (ts.cpu().numpy() == ots).all() and (te.cpu().numpy() == ote).all()
        assert ri.is_contiguous() and ts.is_contiguous() and int(pi[:, 1].sum()) == ri.numel()
        sizes.append(ri.numel())
    assert sizes[1] > 2 * sizes[0] and sizes[2] < sizes[1] and sizes[4] == 0


def test_speculative_expansion_capacity_and_oracle(dev, oracle):
    """The sampler launches the expansion of the run records into arrays sized from the PREVIOUS batch of the same shape,
    before this batch's total has reached the host (grid.py: _SPEC_CAPACITY): whatever capacity was remembered -- equal to
    the total, just below it, far above it -- the results are those of the plain order (read the total, then allocate),
    and the oracle's.  Cases: one level with in-kernel intersection, nested levels, ray counts that are not multiples of
    64 or 256, speckled grids (more run records per ray than MAX_RUNS: serial fill), a mask with a step limit, per-ray near
    planes, an empty grid."""
    from nerfacc_amd import grid as G
    rng = np.random.default_rng(77)
    cases = [  # (R, res, levels, occupancy kind, step, masked)
        (70_001, 64, 1, "shell", 4e-3, False), (4097, 32, 3, 0.3, 0.01, False), (1000, 48, 1, "checker", 0.004, False),
        (33_333, 32, 2, 0.5, 0.02, True), (63, 16, 1, 0.2, 0.01, False), (20_000, 40, 1, 0.0, 0.01, False),
        (50_000, 128, 1, 0.1, 2 * 3 ** 0.5 / 1024, False)]
    import bench
    saved = G.SPECULATE
    try:
        for R, res, levels, occ, step, masked in cases:
            if occ == "shell":
                b = bench.make_grid(res, "shell10")
            elif occ == "checker":
                b = np.broadcast_to((np.indices((res,) * 3).sum(0) % 2 == 0), (levels, res, res, res)).copy()
            else:
                b = rng.random((levels, res, res, res)) < occ
            o = (rng.random((R, 3)) * 3 - 1.5).astype(np.float32)
            d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
            est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=levels).to(dev)
            ab = est.aabbs
            near = T((rng.random(R) * 0.3).astype(np.float32), dev) if levels == 3 else torch.zeros(R, device=dev)
            far = torch.full((R,), 1e10, device=dev)
            kw = dict(return_terminate=True, near_hint=None if levels == 3 else 0.0)
            if masked:
                kw.update(rays_mask=T(rng.random(R) < 0.6, dev), traverse_steps_limit=11)
            args = (T(o, dev), T(d, dev), T(b, dev), ab, near, far, step, 0.0)
            G.SPECULATE = False
            ref = G._traverse_samples(*args, **kw)
            G.SPECULATE = True
            total = ref[0].numel()
            key = (R, dev.index)
            for cap in (total, ((total + 4095) // 4096) * 4096 + 8192, max(total - 1, 1), total * 3 + 100):
                G._SPEC_CAPACITY[key] = max(cap, 1)
                got = G._traverse_samples(*args, **kw)
                assert all(torch.equal(x, y) for x, y in zip(ref, got)), (R, res, levels, occ, cap, total)
                assert got[0].is_contiguous() and got[1].is_contiguous()
            if not masked and levels != 3:
                ori, ots, ote = oracle.occgrid_sampling(o, d, b, ab.cpu().numpy(), render_step_size=step)
                assert (ref[0].cpu().numpy() == ori).all() and (ref[1].cpu().numpy() == ots).all() and (ref[2].cpu().numpy() == ote).all()
    finally:
        G.SPECULATE = saved


def test_speculative_compaction_capacity(dev):
    """The sampler launches the compaction into arrays sized from the PREVIOUS batch's kept fraction before the host has read
    this batch's size (estimators/occ_grid.py: _compact): batches that keep more than that capacity, fewer, everything
    (no compaction at all) and nothing must return exactly what the read-then-compact order returns."""
    from nerfacc_amd.estimators import occ_grid as OG
    import bench
    rng = np.random.default_rng(8)
    R = 20_000
    o = (rng.random((R, 3)) * 3 - 1.5).astype(np.float32)
    d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=64).to(dev)
    est.binaries = T(bench.make_grid(64, "shell10"), dev)
    est.occs = est.binaries.reshape(-1).float()
    ro, rd = T(o, dev), T(d, dev)
    saved = OG._SPECULATE_COMPACTION
    try:
        fracs = []
        for scale in (30.0, 3.0, 300.0, 0.0, 100.0, 1e6, 30.0):     # kept fraction goes up (over capacity), down, to 1, ~0, back
            fn = lambda ts, te, ri: torch.full_like(ts, scale)
            OG._SPECULATE_COMPACTION = False
            ref = est.sampling(ro, rd, sigma_fn=fn, render_step_size=5e-3, early_stop_eps=1e-2)
            OG._SPECULATE_COMPACTION = True
            got = est.sampling(ro, rd, sigma_fn=fn, render_step_size=5e-3, early_stop_eps=1e-2)
            assert all(torch.equal(x, y) for x, y in zip(ref, got)) and got[0].is_contiguous(), scale
            fracs.append(got[0].numel())
        assert fracs[1] > 1.5 * fracs[0] and fracs[2] < fracs[0] and fracs[3] == max(fracs) and fracs[5] < fracs[4]
    finally:
        OG._SPECULATE_COMPACTION = saved


def test_walk_survives_degenerate_rays(dev, oracle):
    """Rays the reference would read out of bounds or spin on (NaN / inf / zero directions, origins 1e30 away, zero-length
    spans, a near plane beyond the far plane): the walk must terminate, touch nothing outside its buffers, give no samples
    to rays without a valid span -- and leave the well-formed rays of the same batch bit-exact."""
    rng = np.random.default_rng(41)
    R, res, step = 2048, 32, 0.01
    o = (rng.random((R, 3)) * 3 - 1.5).astype(np.float32)
    d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    good = np.ones(R, bool)
    bad = rng.choice(R, 400, replace=False)
    good[bad] = False
    d[bad[:50]] = np.nan
    d[bad[50:100], 0] = np.inf
    d[bad[100:150]] = 0.0
    o[bad[150:200]] = 1e30
    o[bad[200:250], 1] = np.nan
    d[bad[250:300]] = np.array([0.0, 0.0, 1.0], np.float32); o[bad[250:300]] = np.array([1.0, 1.0, -3.0], np.float32)   # along an edge
    o[bad[300:350]] = -1e-30; d[bad[300:350]] = 1e-30                                                                   # denormal-ish
    d[bad[350:400]] *= -np.float32(0.0)                                                                                   # signed zeros
    b = rng.random((2, res, res, res)) < 0.3
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=2).to(dev)
    est.binaries = T(b, dev)
    ab = est.aabbs.cpu().numpy()
    ri, ts, te = est.sampling(T(o, dev), T(d, dev), render_step_size=step, far_plane=8.0)
    torch.cuda.synchronize()
    assert ri.numel() > 0 and (ri[1:] >= ri[:-1]).all() and ri.min() >= 0 and ri.max() < R
    assert torch.isfinite(ts).all() and torch.isfinite(te).all() and (te > ts).all()
    # the well-formed rays are exactly the oracle's
    ori, ots, ote = oracle.occgrid_sampling(o[good], d[good], b, ab, render_step_size=step, far_plane=8.0)
    idx = np.flatnonzero(good)
    keep = np.isin(ri.cpu().numpy(), idx)
    remap = np.full(R, -1); remap[idx] = np.arange(idx.size)
    assert (remap[ri.cpu().numpy()[keep]] == ori).all() and (ts.cpu().numpy()[keep] == ots).all() and (te.cpu().numpy()[keep] == ote).all()
    # rays without any finite geometry get nothing
    cnt = torch.bincount(ri, minlength=R).cpu().numpy()
    assert cnt[bad[:50]].sum() == 0 and cnt[bad[50:100]].sum() == 0 and cnt[bad[200:250]].sum() == 0
    # the same rule on the serial kernels (cone angle) and in the API's traverse_grids
    ric, tsc, tec = est.sampling(T(o, dev), T(d, dev), render_step_size=step, far_plane=8.0, cone_angle=0.01)
    cntc = torch.bincount(ric, minlength=R).cpu().numpy()
    assert cntc[bad[:50]].sum() == 0 and cntc[bad[200:250]].sum() == 0 and torch.isfinite(tsc).all()
    iv, sm, _ = na.traverse_grids(T(o, dev), T(d, dev), T(b, dev), T(ab, dev), step_size=step,
                                  far_planes=torch.full((R,), 8.0, device=dev))
    assert int(sm.packed_info[:, 1][torch.from_numpy(bad[:50]).to(dev)].sum()) == 0
    # near plane beyond the far plane / zero-length spans
    r0, t0, t1 = est.sampling(T(o, dev), T(d, dev), render_step_size=step, near_plane=5.0, far_plane=1.0)
    assert r0.numel() == 0


def test_propnet_step_captures_into_a_hipgraph():
    """the fixed-shape PropNet path (sampling level loop, batched transmittance, proposal loss and their
    backward) contains no host synchronisation: it captures into a torch.cuda.CUDAGraph (hipGraph) and the replay is
    bit-identical to the eager step.  Runs in a child process (scripts/graph_bisect.py), as a capture that fails takes its
    process down with it."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "graph_bisect.py")
    for piece in ("sampling_loss_grad", "transmittance_fwd_bwd", "pdf_loss_bwd"):
        r = subprocess.run([sys.executable, script, piece], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, GB_R="8192", GB_EAGER_FIRST="1"))
        assert r.returncode == 0 and f"OK {piece} True" in r.stdout, (piece, r.returncode, r.stdout[-300:], r.stderr[-300:])



def test_captured_propnet_step_replays_identically(dev):
    """PropNetEstimator.capture: the fixed-shape step (level loop, batched transmittance, proposal loss, its gradient) as one
    hipGraph launch (nerfacc_amd.CapturedStep) -- every replay returns exactly the eager step's outputs, also after the
    tensors the step reads were rewritten in place.  In a child process: a capture that fails takes its process down."""
    import subprocess
    import sys
    code = r"""
import sys, torch
sys.path.insert(0, %r)
import nerfacc_amd as na
dev = torch.device("cuda:0")
R = 4096
p = torch.nn.Parameter(torch.tensor([3.0, 4.0], device=dev))
shift = torch.zeros(1, device=dev)
est = na.PropNetEstimator().to(dev)
prop = lambda ts, te: torch.exp(-((ts + te) * 0.5 - p[1] - shift) ** 2) * p[0]
fine = lambda ts, te: torch.exp(-((ts + te) * 0.5 - 4.0) ** 2 * 2.0) * 5.0
def step():
    ts, te = est.sampling([prop, prop], [64, 64], 16, R, 2.0, 6.0, sampling_type="uniform", stratified=False, requires_grad=True)
    trans, _ = na.render_transmittance_from_density(ts, te, fine(ts, te))
    loss = est.compute_loss(trans)
    return ts, te, loss, torch.autograd.grad(loss, [p])[0]
captured = est.capture(step)
ok = True
for k in range(3):
    shift.fill_(0.1 * k)                      # new data written INTO the tensors the step reads
    got = [t.clone() for t in captured()]
    want = step()
    ok = ok and all(torch.equal(a, b) for a, b in zip(got, want))
print("OK captured", ok, captured.replays)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK captured True 3" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-500:])


def test_results_do_not_depend_on_the_tiling():
    """The packed ops (rendering forward / backward, visibility, weights, accumulation, scans) on twelve random ragged
    batches -- empty rays, runs of tiny rays, rays of thousands of samples -- give the same bits whatever the tile size:
    each size runs in a child process (scripts/tiling_invariance.py hands NFA_SEG_TILE to nfa_set_tuning)
    and the digests of all outputs are compared."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "tiling_invariance.py")
    digests = []
    for tile in ("256", "1024", "3072"):
        r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, NFA_SEG_TILE=tile))
        assert r.returncode == 0 and "digest " in r.stdout, (tile, r.returncode, r.stdout[-300:], r.stderr[-300:])
        digests.append(r.stdout.strip().splitlines()[-1])
    assert digests[0] == digests[1] == digests[2], digests


def test_bench_under_torchrun_exercises_rccl(dev):
    """BASELINE cfg 4's code path in front of the driver: a fresh child process under `torch.distributed.run
    --nproc-per-node 1` (the launcher starts before anything in the child touches the GPU) runs bench.py on the shared
    256^3 grid -- RCCL init with device_id, the bit-packed grid broadcast from rank 0, the SUM all-reduce of the parameter
    gradient in every step, barrier and MAX-over-ranks timing -- and prints the one JSON line."""
    import hashlib
    import json
    import socket
    import subprocess
    import sys
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--res", "256", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-extras", "--no-pipelined", "--no-kernel-timing"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "rays/s" and d["value"] > 1e6 and d["scaling"] == "weak"
    dist = d["distributed"]
    assert dist["backend"] == "nccl" and dist["world_size"] == 1
    want = hashlib.sha256(np.packbits(bench.make_grid(256, "shell10").reshape(-1)).tobytes()).hexdigest()
    assert dist["shared_grid_sha256"] == want and dist["shared_grid_bytes"] == 256 ** 3 // 8
    assert dist["grad_finite"] and len(dist["grad_after_allreduce"]) == 2 and dist["grad_allreduces"] >= 4
    assert any(abs(x) > 0 for x in dist["grad_after_allreduce"])


def test_cone_walk_record_arena(dev, oracle):
    """Rays with more run records than their MAX_RUNS slots keep MAX_RUNS - 1 of them and a sentinel, the others go to the
    arena and are expanded from there (csrc/walk.hip: ConeParams::arena, nfa_expand_cone_arena) -- no second walk.  With 4
    slots per ray nearly every ray uses the arena; with a 16-entry arena most rays find it full and take the serial fill
    pass after all; results: the oracle's, bit for bit, in every combination, and the form without an arena."""
    from nerfacc_amd import grid as G
    rng = np.random.default_rng(11)
    saved = (G.MAX_RUNS, G.CONE_ARENA, G.CONE_ARENA_MIN)
    try:
        for levels, res, occ, R, step, cone, limit in ((2, 32, 0.3, 6000, 6e-3, 0.01, None), (1, 48, 0.5, 3000, 3e-3, 0.004, None),
                                                        (3, 24, 0.2, 4000, 5e-3, 0.02, 30)):
            est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=levels).to(dev)
            b = rng.random((levels, res, res, res)) < occ
            o = (rng.random((R, 3)).astype(np.float32) - 0.5) * 1.5
            d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
            near = np.full(R, 0.05, np.float32); far = np.full(R, 1e10, np.float32)
            mask = (rng.random(R) < 0.7) if limit else None
            kw = dict(rays_mask=None if mask is None else T(mask, dev), traverse_steps_limit=limit, near_hint=0.05)
            args = (T(o, dev), T(d, dev), T(b, dev), est.aabbs, T(near, dev), T(far, dev), step, cone)
            outs = {}
            for name, (max_runs, arena, amin) in {"32 slots": (32, True, 4096), "4 slots + arena": (4, True, 1 << 20),
                                                  "4 slots, arena of 16": (4, True, 16), "4 slots, no arena": (4, False, 4096),
                                                  "2 slots + arena": (2, True, 1 << 20)}.items():
                G.MAX_RUNS, G.CONE_ARENA, G.CONE_ARENA_MIN = max_runs, arena, amin
                calls = []
                orig = G.B.call
                G.B.call = lambda n, *a, **k: (calls.append(n), orig(n, *a, **k))[1]
                try:
                    outs[name] = G._traverse_samples(*args, **kw)
                finally:
                    G.B.call = orig
                if name == "4 slots + arena":
                    assert "nfa_expand_cone_arena" in calls and "nfa_traverse_grids" not in calls, calls
                if name == "4 slots, no arena":
                    assert "nfa_expand_cone_arena" not in calls and "nfa_traverse_grids" in calls, calls
            ref = outs["32 slots"]
            assert ref[0].numel() > 2000
            for name, got in outs.items():
                assert all(torch.equal(x, y) for x, y in zip(ref, got)), (name, levels, res)
            ab = est.aabbs.cpu().numpy()
            if limit is None:
                riv, rsm, _ = oracle.traverse_grids(o, d, b, ab, near_planes=near, far_planes=far, step_size=step, cone_angle=cone)
                keep = np.ones(rsm["ray_indices"].shape[0], bool)
            else:
                riv, rsm, _ = oracle.traverse_grids(o, d, b, ab, near_planes=near, far_planes=far, step_size=step, cone_angle=cone,
                                                    traverse_steps_limit=limit, over_allocate=True, rays_mask=mask)
                keep = rsm["is_valid"]
            assert (ref[0].cpu().numpy() == rsm["ray_indices"][keep]).all()
            assert (ref[1].cpu().numpy() == riv["vals"][riv["is_left"]]).all() and (ref[2].cpu().numpy() == riv["vals"][riv["is_right"]]).all()
    finally:
        G.MAX_RUNS, G.CONE_ARENA, G.CONE_ARENA_MIN = saved


@pytest.mark.parametrize("levels,alpha_thre,graph", [(1, 0.0, True), (2, 0.0, True), (1, 2e-3, True), (1, 0.0, False)])
def test_padded_test_mode_loop_equals_the_exact_one(dev, levels, alpha_thre, graph):
    """nerfacc_amd.marching.PaddedTestModeLoop -- the test-mode loop with fixed shapes, the schedule on the device and one
    iteration replayed as a hipGraph, no host read inside -- gives the image and the sample count of the exact-shape loop
    (same kernels on the same values: bit for bit), on one and two levels, with an alpha threshold, and replayed a second time."""
    from nerfacc_amd.marching import render_rays_test_mode, PaddedTestModeLoop
    rng = np.random.default_rng(23)
    n, res, step = 6000, 48, 6e-3
    o = (rng.random((n, 3)).astype(np.float32) - 0.5) * 3.0
    d = rng.standard_normal((n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    b = rng.random((levels, res, res, res)) < 0.25
    est = na.OccGridEstimator([-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=levels).to(dev)
    est.binaries = T(b, dev)
    bk = T(np.array([0.2, 0.4, 0.6], np.float32), dev)

    def field_t(ts, te, ri):
        tm = (ts + te) * 0.5
        return (torch.stack([0.5 + 0.5 * torch.cos(tm), (ri % 7).float() / 7.0, torch.full_like(tm, 0.3)], -1),
                25.0 * (0.5 + 0.5 * torch.sin(9.0 * tm)))

    kw = dict(near_plane=0.05, far_plane=1e10, render_step_size=step, alpha_thre=alpha_thre, early_stop_eps=1e-3)
    import nerfacc_amd.marching as marching
    shapes = []

    def field_exact(ts, te, ri):   # the exact-shape contract: the callback sees the iteration's samples and nothing else
        shapes.append((ts.shape[0], te.shape[0], ri.shape[0]))
        return field_t(ts, te, ri)

    marching.ONE_READ = False      # rounds 1-3's loop (two host reads per iteration)
    try:
        want = render_rays_test_mode(600, field_exact, est, T(o, dev), T(d, dev), render_bkgd=bk, **kw)
    finally:
        marching.ONE_READ = True
    shapes_two_reads, shapes = shapes, []
    one = render_rays_test_mode(600, field_exact, est, T(o, dev), T(d, dev), render_bkgd=bk, **kw)   # one read per iteration (the default)
    assert one[3] == want[3] and all(torch.equal(a, w) for a, w in zip(one[:3], want[:3]))
    assert shapes == shapes_two_reads and len(shapes) >= 10 and all(a == b == c for a, b, c in shapes)
    loop = PaddedTestModeLoop(600, field_t, est, T(o, dev), T(d, dev), kw["near_plane"], kw["far_plane"], step, alpha_thre, 1e-3,
                              use_graph=graph)
    for rep in range(2):
        got = loop.render(bk)
        assert got[3] == want[3] and want[3] > 20000, (rep, got[3], want[3])
        for a, w in zip(got[:3], want[:3]):
            assert torch.equal(a, w), rep
        assert loop.iterations_run >= 10 and loop.iterations_queued <= loop.iterations_run + 3 * loop.check_every
    if graph:   # the public switch
        got = render_rays_test_mode(600, field_t, est, T(o, dev), T(d, dev), render_bkgd=bk, padded=True, **kw)
        assert got[3] == want[3] and all(torch.equal(a, w) for a, w in zip(got[:3], want[:3]))


def test_cell_selection_peak_memory(dev):
    """OccGridEstimator._sample_uniform_and_occupied_cells keeps its temporaries small on large grids (the one-read
    form's all-level int64 prefix sums took gigabytes at 4 x 512^3): int32 prefix sums up to 2^27 cells (4 x 256^3: measured
    525 MB), beyond that the reference's per-level expressions (4 x 512^3: measured 122 MB).  Both return, per level, the
    kept uniform draws followed by the occupied cells, all in range."""
    for res, limit_mb in ((256, 800), (512, 400)):
        est = na.OccGridEstimator([-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=4).to(dev)
        est.binaries[:, :, :, ::10] = True
        est.occs.fill_(0.5)
        n = 1 << 18
        torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); base = torch.cuda.memory_allocated()
        out = est._sample_uniform_and_occupied_cells(n)
        torch.cuda.synchronize()
        peak_mb = (torch.cuda.max_memory_allocated() - base) / 2 ** 20
        assert peak_mb < limit_mb, (res, peak_mb)
        assert len(out) == 4
        for cells in out:
            assert cells.dtype == torch.int64 and cells.numel() == 2 * n            # every draw is visible (occs >= 0), n occupied cells drawn
            assert int(cells.min()) >= 0 and int(cells.max()) < res ** 3
            occ = cells[n:]
            assert bool(((occ % res) % 10 == 0).all())                               # the occupied half really is occupied (z % 10 == 0)
        del est, out
        torch.cuda.empty_cache()
