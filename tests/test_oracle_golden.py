"""CPU: the oracle (oracle/nerfacc_oracle.c) against the committed golden fixtures (outputs of
the imported reference, oracle/gen_golden.py) and against the reference's own hard-coded test
vectors."""
import hashlib

import numpy as np
import torch

from conftest import assert_close, load_golden, seeded_case


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_reference_known_answers(oracle):
    O = oracle
    # tests/test_pack.py:11-18
    assert O.pack_info([0, 2, 2, 2, 2], 3).tolist() == [[0, 1], [1, 0], [1, 4]]
    pi = np.array([[0, 1], [1, 0], [1, 4]])
    alphas = np.array([0.4, 0.3, 0.8, 0.8, 0.5], np.float32)
    # tests/test_rendering.py:11-34
    assert O.render_visibility_from_alpha(alphas, pi, 0.03, 0.0).tolist() == [True, True, True, True, False]
    assert O.render_visibility_from_alpha(alphas, pi, 0.05, 0.35).tolist() == [True, False, True, True, False]
    # tests/test_rendering.py:41-57
    w, _ = O.render_weight_from_alpha(alphas, pi)
    assert np.allclose(w, [0.4, 0.3, 0.7 * 0.8, 0.14 * 0.8, 0.028 * 0.5])
    # tests/test_rendering.py:117-133 (test_grads), t_ends = t_starts + 1
    sig = np.array([0.4, 0.8, 0.1, 0.8, 0.1], np.float32)
    ts = np.random.default_rng(0).random(5).astype(np.float32)
    w, _, _ = O.render_weight_from_density(ts, ts + 1, sig, pi)
    assert np.allclose(w, [0.3297, 0.5507, 0.0428, 0.2239, 0.0174], atol=1e-4)
    g = O.render_weight_from_density_backward(ts, ts + 1, sig, pi, np.ones(5))
    assert np.allclose(g, [0.6703, 0.1653, 0.1653, 0.1653, 0.1653], atol=1e-4)
    # docstrings scan.py:36-39,78-81,127-130,170-173
    x = np.arange(1, 10, dtype=np.float32)
    p3 = np.array([[0, 2], [2, 3], [5, 4]])
    assert O.inclusive_sum(x, p3).tolist() == [1, 3, 3, 7, 12, 6, 13, 21, 30]
    assert O.exclusive_sum(x, p3).tolist() == [0, 1, 0, 3, 7, 0, 6, 13, 21]
    assert O.inclusive_prod(x, p3).tolist() == [1, 2, 3, 12, 60, 6, 42, 336, 3024]
    assert O.exclusive_prod(x, p3).tolist() == [1, 1, 1, 3, 12, 1, 6, 42, 336]
    # docstrings volrend.py:192-195, 298-302
    a7 = np.array([0.4, 0.8, 0.1, 0.8, 0.1, 0.0, 0.9], np.float32)
    p7 = O.pack_info([0, 0, 0, 1, 1, 2, 2], 3)
    assert np.allclose(O.render_transmittance_from_alpha(a7, p7), [1.0, 0.6, 0.12, 1.0, 0.2, 1.0, 1.0])
    assert np.allclose(O.render_weight_from_alpha(a7, p7)[0], [0.4, 0.48, 0.012, 0.8, 0.02, 0.0, 0.9])
    assert O.render_visibility_from_alpha(a7, p7, 0.3, 0.2).tolist() == [True, True, False, True, False, False, True]
    # Philox4x32-10 known-answer vectors (Random123 kat_vectors)
    assert [hex(v) for v in O.philox4x32_10([0] * 4, [0] * 2)] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(v) for v in O.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2)] == \
        ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(v) for v in O.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_ragged_packed_vs_reference(oracle):
    O, g = oracle, load_golden("ragged_packed")
    pi = g["packed_info"]
    for kind, inp in (("inclusive_sum", g["x"]), ("exclusive_sum", g["x"]), ("inclusive_prod", g["xp"]),
                      ("exclusive_prod", g["xp"])):
        y = O.packed_scan(kind, inp, pi)
        assert_close(y, g[kind], atol=1e-5, rtol=2e-5, what=kind)
        gi = O.sum_backward(kind, g["g"], pi) if kind.endswith("sum") else O.prod_backward(kind, inp, y, g["g"], pi)
        assert_close(gi, g[kind + "_grad"], atol=2e-5, rtol=1e-4, what=kind + " grad")
    w, t, a = O.render_weight_from_density(g["ts"], g["te"], g["sig"], pi)
    assert_close(w, g["rwd_w"]); assert_close(t, g["rwd_t"]); assert_close(a, g["rwd_a"])
    gs = O.render_weight_from_density_backward(g["ts"], g["te"], g["sig"], pi, g["gw"], g["gt"], g["ga"])
    assert_close(gs, g["rwd_gsig"], atol=2e-5, rtol=1e-4)
    w2, t2, _ = O.render_weight_from_density(g["ts"], g["te"], g["sig"], pi, prefix_trans=g["pref"])
    assert_close(w2, g["rwd_pref_w"]); assert_close(t2, g["rwd_pref_t"])
    wa, ta = O.render_weight_from_alpha(g["alph"], pi)
    assert_close(wa, g["rwa_w"]); assert_close(ta, g["rwa_t"])
    vd = O.render_visibility_from_density(g["ts"], g["te"], g["sig"], pi, float(g["eps_t"]), float(g["thre"]))
    va = O.render_visibility_from_alpha(g["alph"], pi, float(g["eps_t"]), float(g["thre"]))
    assert ((vd == g["vis_d"]) | g["guard_d"]).all() and ((va == g["vis_a"]) | g["guard_a"]).all()
    n_rays = pi.shape[0]
    assert_close(O.accumulate_along_rays(g["rwd_w"], g["rgb"], g["ray_indices"], n_rays), g["acc_rgb"], atol=1e-5)
    c, o, d, _ = O.rendering(g["ts"], g["te"], g["ray_indices"], n_rays, g["rgb"], sigmas=g["sig"], render_bkgd=g["bkgd"])
    assert_close(c, g["rend_colors"], atol=1e-5); assert_close(d, g["rend_depths"], atol=1e-5, rtol=1e-4)
    assert (O.pack_info(g["ray_indices"], n_rays) == pi).all()


def test_ray_aabb_vs_reference_twin(oracle):
    g = load_golden("ray_aabb")
    tm, tM, hit = oracle.ray_aabb_intersect(g["rays_o"], g["rays_d"], g["aabbs"])
    assert (hit == g["hits"]).all()
    assert np.allclose(tm, g["t_mins"]) and np.allclose(tM, g["t_maxs"])  # tests/test_grid.py:25-27


def test_pdf_vs_reference_twin(oracle):
    O, g = oracle, load_golden("pdf")
    for tag in "abc":
        iv, sm = O.importance_sampling(g[f"{tag}_vals"], g[f"{tag}_cdfs"], int(g[f"{tag}_S"]), False)
        assert_close(iv, g[f"{tag}_twin_edges"], atol=1e-4, rtol=0)  # tests/test_pdf.py:93-94
        assert_close(sm, g[f"{tag}_twin_centres"], atol=1e-4, rtol=0)
        assert (iv == g[f"{tag}_oracle_edges"]).all() and (sm == g[f"{tag}_oracle_centres"]).all()
    il, ir = O.searchsorted(g["loss_k_vals"], g["loss_q_vals"])
    ref = torch.clamp(torch.searchsorted(torch.from_numpy(g["loss_k_vals"]), torch.from_numpy(g["loss_q_vals"]), right=True),
                      0, g["loss_k_vals"].shape[-1] - 1).numpy()
    assert (ir == ref).all()  # tests/test_pdf.py:57-62
    assert (il == g["loss_ids_left"]).all()


def test_traversal_golden(oracle):
    O, g = oracle, load_golden("traversal")
    binaries = np.unpackbits(g["a_binaries"]).astype(bool).reshape(4, 32, 32, 32)
    iv, sm, term = O.traverse_grids(g["a_rays_o"][:8], g["a_rays_d"][:8], binaries, g["a_aabbs"])
    assert (iv["vals"] == g["a8_iv_vals"]).all() and (iv["is_left"] == g["a8_iv_left"]).all()
    assert (iv["is_right"] == g["a8_iv_right"]).all() and (iv["packed_info"] == g["a8_iv_packed"]).all()
    assert (sm["packed_info"] == g["a8_sm_packed"]).all() and (term == g["a8_term"]).all()
    for tag in ("cfg1", "cone", "percell"):
        o, d, b, ab, nearp, step, cone = seeded_case(g[f"{tag}_params"])
        iv, sm, term = O.traverse_grids(o, d, b, ab, near_planes=nearp, step_size=step, cone_angle=cone)
        assert len(sm["vals"]) == int(g[f"{tag}_M"]) and len(iv["vals"]) == int(g[f"{tag}_E"])
        assert sha(sm["packed_info"]) == str(g[f"{tag}_sm_cnts_sha"])
        assert sha(iv["vals"]) == str(g[f"{tag}_iv_vals_sha"])
        assert sha(term) == str(g[f"{tag}_term_sha"])


def _propnet_field(off):
    return lambda ts, te: (np.exp(-((ts + te) * np.float32(0.5) - np.float32(4.0) - off) ** 2 * np.float32(2.0))
                           * np.float32(3.0) + np.float32(0.05)).astype(np.float32)


def test_propnet_sampling_vs_reference_fixture(oracle):
    """oracle.propnet_sampling against the outputs of the reference's own PropNetEstimator.sampling loop
    (estimators/prop_net.py:38-129; oracle/gen_golden.py: propnet_fixtures)."""
    g = load_golden("propnet")
    fn = _propnet_field(g["off"])
    n = g["off"].shape[0]
    for tag, kind in (("u", "uniform"), ("l", "lindisp")):
        props = [int(v) for v in g[f"{tag}_props"]]
        ts, te, levels = oracle.propnet_sampling([fn] * len(props), props, int(g[f"{tag}_final"]), n, 2.0, 6.0, sampling_type=kind)
        assert_close(ts, g[f"{tag}_t_starts"], atol=2e-6)
        assert_close(te, g[f"{tag}_t_ends"], atol=2e-6)
        assert_close(levels[0][1], g[f"{tag}_cdfs0"], atol=1e-6)


def test_grid_maintenance_vs_reference_fixture(oracle):
    """oracle.grid_rebinarize / mark_invisible_cells against what the reference's OccGridEstimator left behind
    (oracle/gen_golden.py: occgrid_fixtures; estimators/occ_grid.py:262-332, 403-404)."""
    g = load_golden("occgrid")
    shape = tuple(g["sd_binaries"].shape)
    for tag in ("warm", "samp"):
        b, thre = oracle.grid_rebinarize(g[f"{tag}_occs"], shape, 0.02)
        ref = np.unpackbits(g[f"{tag}_binaries"])[:b.size].astype(bool).reshape(shape)
        assert int((b != ref).sum()) <= 2 and abs(float(thre) - float(g[f"{tag}_thre"])) < 1e-9
    res = [int(v) for v in g["sd_resolution"]]
    marked = oracle.mark_invisible_cells(np.zeros_like(g["occs_marked"]), res, g["sd_aabbs"], g["K"], g["c2w"], int(g["W"]), int(g["H"]),
                                         float(g["near"]))
    assert int((marked != g["occs_marked"]).sum()) <= 8


def test_oracle_test_mode_loop_consistent_with_one_shot_rendering(oracle):
    """The oracle's restatement of the test-mode marching loop (examples/utils.py:252-425; the reference's own harness
    needs its CUDA extension, so this row is pinned through the pinned pieces it is composed of and through the
    reference's own property, tests/test_grid.py:72-131: marching in chunks == marching in one go).  With early
    termination and the opacity threshold off, the loop must visit exactly the one-shot traversal's samples and blend the
    same image (the chunked prefix transmittance is the one-shot transmittance); with them on, rays stop early and the
    image stays within the threshold's worth of the full one."""
    rng = np.random.default_rng(3)
    n, res, step = 300, 24, 0.02
    o = (rng.random((n, 3)).astype(np.float32) - 0.5) * 3.0
    d = rng.standard_normal((n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    b = rng.random((2, res, res, res)) < 0.3
    ab = np.array([[-1, -1, -1, 1, 1, 1], [-2, -2, -2, 2, 2, 2]], np.float32)
    bk = np.array([0.2, 0.4, 0.6], np.float32)

    def field(ts, te, ri):
        tm = (ts + te) * np.float32(0.5)
        sig = (np.float32(3.0) * (np.float32(0.5) + np.float32(0.5) * np.sin(np.float32(5.0) * tm))).astype(np.float32)
        return np.stack([tm * np.float32(0.1), np.full_like(tm, 0.5), (ri % 3).astype(np.float32) / 3], -1).astype(np.float32), sig

    for cone in (0.0, 0.01):
        rgb, opa, dep, total, info = oracle.test_mode_loop(100000, field, o, d, b, ab, near_plane=0.05, render_step_size=step,
                                                           render_bkgd=bk, cone_angle=cone, early_stop_eps=0.0)
        ri, ts, te = oracle.occgrid_sampling(o, d, b, ab, near_plane=0.05, render_step_size=step, cone_angle=cone)
        assert total == ri.size and (info["samples_per_ray"] == np.bincount(ri, minlength=n)).all() and total > 5000
        rgbs, sig = field(ts, te, ri)
        c1, o1, d1, _ = oracle.rendering(ts, te, ri, n, rgbs, sigmas=sig, render_bkgd=bk)
        assert np.allclose(rgb, c1, atol=2e-5) and np.allclose(opa, o1, atol=2e-5) and np.allclose(dep, d1, atol=2e-4)
        dense = lambda ts, te, ri: (field(ts, te, ri)[0], field(ts, te, ri)[1] * np.float32(40.0))
        rgb_f, opa_f, _, total_f, _ = oracle.test_mode_loop(100000, dense, o, d, b, ab, near_plane=0.05, render_step_size=step,
                                                            render_bkgd=bk, cone_angle=cone, early_stop_eps=0.0)
        rgb_e, opa_e, _, total_e, _ = oracle.test_mode_loop(100000, dense, o, d, b, ab, near_plane=0.05, render_step_size=step,
                                                            render_bkgd=bk, cone_angle=cone, early_stop_eps=0.05)
        assert total_e < total_f and np.abs(rgb_e - rgb_f).max() < 0.06 and (opa_e <= opa_f + 1e-6).all()


def test_threaded_bench_step_agrees_with_the_pinned_composition(oracle):
    """oracle.bench_step (bench.py's timed CPU baseline: every stage an OpenMP loop over rays, per-ray serial scans) against
    the composition of the pinned oracle functions (bench._oracle_step): the same samples -- traversal bit for bit,
    visibility identical outside the threshold's guard band -- colours and density gradients within 1e-5."""
    import bench
    o, d = bench.make_rays(64 * 64, "image")
    b = bench.make_grid(32, "shell10")
    aabb = np.array([[-1, -1, -1, 1, 1, 1]], np.float32)
    step = 2 * 3 ** 0.5 / 256
    for scale in (1.0, 16.0):
        kept, full, colors, gsig, sig = bench._oracle_step(oracle, o, d, b, aabb, step, scale)
        (ri, ts, te), M, col2, g2 = oracle.bench_step(o, d, b, aabb, step, scale)
        assert M == full[0].size and M > 20000
        tr, _ = oracle.render_transmittance_from_density(full[1], full[2], sig(full[1], full[2], full[0]), full[3])
        guard = np.abs(tr - np.float32(1e-4)) < 1e-6
        if not guard.any():
            assert (ri == kept[0]).all() and (ts == kept[1]).all() and (te == kept[2]).all()
            assert np.allclose(col2, colors, atol=1e-5) and np.allclose(g2, gsig, atol=1e-5, rtol=1e-4)
        else:
            assert abs(ri.size - kept[0].size) <= int(guard.sum())
        if scale == 16.0:
            assert kept[0].size < M          # early termination bites


def test_proposal_loss_backward_vs_reference_autograd(oracle):
    """oracle.pdf_loss_batched / pdf_loss_batched_backward / density_cdf_backward (the backward half of BASELINE cfg 3's CPU
    baseline) against torch autograd of the reference's expressions (prop_net.py:254-255, :113; volrend.py:245-264),
    recorded by oracle/gen_golden.py."""
    g = load_golden("pdf")
    l, saved = oracle.pdf_loss_batched(g["loss_q_vals"], g["loss_q_cdfs"], g["loss_k_vals"], g["loss_k_cdfs"])
    assert_close(l, g["lossb_loss"], atol=1e-7, rtol=1e-6)
    assert_close(oracle.pdf_loss_batched_backward(g["lossb_gl"], saved), g["lossb_gk"], atol=1e-6, rtol=1e-5)
    assert_close(oracle.density_cdf_backward(g["cdfb_t0"], g["cdfb_t1"], g["cdfb_sig"], g["cdfb_gc"]), g["cdfb_gsig"], atol=1e-6, rtol=1e-5)
