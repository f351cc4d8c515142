"""CPU: host-side logic of the package -- batched (pure torch) paths, API surface, and the rule
that packed/native ops never silently fall back on the CPU."""
import numpy as np
import pytest
import torch

import nerfacc_amd as na
from conftest import assert_close, load_golden


def test_api_surface_matches_reference():
    """Every public callable of the reference (nerfacc/__init__.py:23-46) and the estimators' methods exist with the
    reference's parameters -- same names, order, kinds and defaults; tests/golden/api_signatures.json was written by
    oracle/gen_golden.py from the imported reference.  Extensions are allowed only as further parameters WITH defaults
    behind the reference's (or keyword-only), so every call the reference accepts means the same here."""
    import inspect
    import json
    import os
    from conftest import ROOT
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "api_signatures.json")))
    assert len(ref) >= 29
    for name, want in ref.items():
        obj = na
        for part in name.split("."):
            assert hasattr(obj, part), name
            obj = getattr(obj, part)
        got = [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)]
               for p in inspect.signature(obj).parameters.values()]
        var_kw = [p for p in want if p[1] == "VAR_KEYWORD"]
        want_fixed = [p for p in want if p[1] != "VAR_KEYWORD"]
        got_fixed = [p for p in got if p[1] != "VAR_KEYWORD"]
        assert got_fixed[:len(want_fixed)] == want_fixed, (name, got_fixed[:len(want_fixed)], want_fixed)
        for extra in got_fixed[len(want_fixed):]:
            assert extra[2] is not None or extra[1] == "KEYWORD_ONLY", (name, extra)
        if var_kw:
            assert any(p[1] == "VAR_KEYWORD" for p in got), name


def test_batched_paths_match_reference_outputs():
    g = load_golden("batched_volrend")
    ts, te = torch.from_numpy(g["ts"]), torch.from_numpy(g["te"])
    sig = torch.from_numpy(g["sig"]).requires_grad_(True)
    w, tr, al = na.render_weight_from_density(ts, te, sig)
    w.sum().backward()
    assert_close(w, g["w"]); assert_close(tr, g["trans"]); assert_close(al, g["alphas"])
    assert_close(sig.grad, g["gsig"], atol=1e-5, rtol=1e-4)
    x = torch.rand(4, 9)
    assert torch.allclose(na.inclusive_sum(x), torch.cumsum(x, -1))
    assert torch.allclose(na.exclusive_prod(x)[:, 1:], torch.cumprod(x, -1)[:, :-1])
    assert torch.allclose(na.accumulate_along_rays(x, x[..., None]), (x * x).sum(-1, keepdim=True))


def test_packed_ops_refuse_cpu_tensors():
    ri = torch.tensor([0, 2, 2, 2, 2])
    with pytest.raises(NotImplementedError):
        na.pack_info(ri, 3)
    x = torch.rand(5)
    with pytest.raises(NotImplementedError):
        na.exclusive_sum(x, torch.tensor([[0, 1], [1, 0], [1, 4]]))
    with pytest.raises(NotImplementedError):
        na.ray_aabb_intersect(torch.rand(4, 3), torch.rand(4, 3), torch.rand(2, 6))
    with pytest.raises(NotImplementedError):
        na.traverse_grids(torch.rand(4, 3), torch.rand(4, 3), torch.ones(1, 2, 2, 2, dtype=torch.bool),
                          torch.tensor([[0., 0, 0, 1, 1, 1]]))


def test_estimator_state_dict_layout():
    est = na.OccGridEstimator([-1, -1, -1, 1, 1, 1], resolution=8, levels=2)
    sd = est.state_dict()
    assert set(sd) == {"resolution", "aabbs", "occs", "binaries"}  # occ_grid.py:67-75
    assert sd["binaries"].dtype == torch.bool and sd["binaries"].shape == (2, 8, 8, 8)
    assert sd["occs"].shape == (2 * 512,) and sd["aabbs"].tolist()[1] == [-2, -2, -2, 2, 2, 2]


def test_mark_invisible_cells_known_counts():
    # tests/test_grid.py:207-233 (pure torch in the reference as well)
    est = na.OccGridEstimator(torch.tensor([-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]), resolution=32, levels=4)
    K = torch.tensor([[[100.0, 0, 50.0], [0, 100.0, 50.0], [0, 0, 1]]])
    pose = torch.tensor([[[-1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, -1.0, 2.5]]])
    est.mark_invisible_cells(K, pose, 100, 100)
    assert int((est.occs == -1).sum()) == 77660
    assert int((est.occs == 0).sum()) == 53412


def test_transform_stot_and_schedule():
    from nerfacc_amd.estimators.prop_net import _transform_stot, get_proposal_requires_grad_fn
    g = load_golden("pdf")
    s = torch.from_numpy(g["stot_s"])
    assert_close(_transform_stot("uniform", s, 2.0, 6.0), g["stot_uniform"])
    assert_close(_transform_stot("lindisp", s, 2.0, 6.0), g["stot_lindisp"])
    fn = get_proposal_requires_grad_fn(target=5.0, num_steps=10)
    fires = [fn(s) for s in range(40)]
    assert fires[0] is False and sum(fires) > 3


def test_guard_band_sampling_checker():
    """oracle/check.py (used by smoke(), bench.py and the GPU tests): identical outputs pass; a sample on the visibility
    threshold may fall on either side; any other difference fails."""
    from oracle import check as OC
    ri = np.array([0, 0, 0, 1, 1, 2], np.int64)
    ts = np.array([0.1, 0.2, 0.3, 0.1, 0.2, 0.5], np.float32)
    te = ts + np.float32(0.1)
    trans = np.array([1.0, 0.5, 1e-4 + 2e-8, 1.0, 0.3, 1.0], np.float32)   # sample 2 sits on the 1e-4 threshold
    vis = trans >= np.float32(1e-4)
    kept = (ri[vis], ts[vis], te[vis])
    full = (ri, ts, te)
    ok, info = OC.compare_sampling(kept, kept, full, trans, None, early_stop_eps=1e-4)
    assert ok and info["identical"]
    drop2 = np.array([True, True, False, True, True, True])
    ok, info = OC.compare_sampling((ri[drop2], ts[drop2], te[drop2]), kept, full, trans, None, early_stop_eps=1e-4)
    assert ok and info["guarded"] == 1 and not info["identical"]
    drop1 = np.array([True, False, True, True, True, True])                  # a sample far from the threshold is missing
    ok, info = OC.compare_sampling((ri[drop1], ts[drop1], te[drop1]), kept, full, trans, None, early_stop_eps=1e-4)
    assert not ok and info["missing"] == 1
    te_bad = te.copy(); te_bad[4] += np.float32(1e-3)
    ok, info = OC.compare_sampling((ri[drop2], ts[drop2], te_bad[drop2]), kept, full, trans, None, early_stop_eps=1e-4)
    assert not ok


def test_reference_state_dict_loads_and_mark_invisible_cells_cpu():
    """A state_dict written by the reference's OccGridEstimator (fixture: names, shapes, dtypes as on disk) loads with
    strict=True, and mark_invisible_cells (torch, ref occ_grid.py:262-332) reproduces the reference's marking."""
    g = load_golden("occgrid")
    res = [int(v) for v in g["sd_resolution"]]
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=2)
    sd = {k: torch.from_numpy(g["sd_" + k]) for k in ("resolution", "aabbs", "occs", "binaries")}
    for k, v in sd.items():
        assert str(v.dtype) == str(g["sd_dtype_" + k]) and est.state_dict()[k].dtype == v.dtype and est.state_dict()[k].shape == v.shape
    est.load_state_dict(sd, strict=True)
    assert list(est.state_dict().keys()) == ["resolution", "aabbs", "occs", "binaries"]
    assert torch.equal(est.binaries, sd["binaries"]) and torch.equal(est.occs, sd["occs"])
    est.occs.zero_()
    est.mark_invisible_cells(torch.from_numpy(g["K"]), torch.from_numpy(g["c2w"]), int(g["W"]), int(g["H"]), near_plane=float(g["near"]))
    assert int((est.occs.numpy() != g["occs_marked"]).sum()) <= 4


def test_cell_selection_without_nonzero_matches_the_reference_expressions():
    """OccGridEstimator._sample_uniform_and_occupied_cells on a device (ref estimators/occ_grid.py:345-366): stable compactions
    by prefix sum + scatter with one host read for all levels.  Same cells in the same order as the reference's boolean
    index + nonzero when every occupied cell is taken (same RNG stream: no selector draws); with more occupied cells than
    n, n draws among the occupied cells."""
    import torch
    import nerfacc_amd as na
    est = na.OccGridEstimator([-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=8, levels=3)
    torch.manual_seed(0)
    occs = torch.rand(3 * 512) - 0.3
    occs[occs < 0] = -1.0                                  # cells no camera sees
    est.occs = occs
    est.binaries = torch.rand(3, 8, 8, 8) < 0.2
    n_occ = est.binaries.view(3, -1).sum(1)
    n = int(n_occ.max()) + 5                               # all occupied cells are taken on every level
    torch.manual_seed(7)
    got = est._sample_cells_one_read(n)
    torch.manual_seed(7)
    for lvl in range(3):
        uni = torch.randint(512, (n,))
        uni = uni[est.occs[lvl * 512 + uni] >= 0.0]
        want = torch.cat([uni, torch.nonzero(est.binaries[lvl].flatten())[:, 0]])
        assert torch.equal(got[lvl], want), lvl
    n = 20                                                   # fewer than the occupied cells: n of them, with replacement
    torch.manual_seed(9)
    got = est._sample_cells_one_read(n)
    for lvl in range(3):
        n_uni = got[lvl].numel() - n
        assert 0 < n_uni <= n and (est.occs[lvl * 512 + got[lvl][:n_uni]] >= 0).all()
        assert est.binaries[lvl].flatten()[got[lvl][n_uni:]].all()
    est.binaries = torch.zeros_like(est.binaries)           # nothing occupied
    assert all(g.numel() <= 5 and g.numel() > 0 for g in est._sample_cells_one_read(5))


def test_walk_limits_are_mirrored_on_the_host():
    import torch
    from nerfacc_amd import grid as G
    ok = lambda shape: G._walk_supported(torch.empty(shape, dtype=torch.bool, device="meta"))
    assert ok((1, 128, 128, 128)) and ok((4, 512, 512, 512)) and ok((8, 512, 512, 512)) and ok((1, 1, 1, 1))
    assert not ok((16, 512, 512, 512)) and not ok((1, 600, 4, 4))
